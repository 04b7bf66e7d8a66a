// xr_dial3.h — XR-Maze v1/v2, LDS form of the frontier router, round 3.  Included by xr_kernels.hip only
// (device code, gfx950 / wave64).  Same results as xr_dial.h / the oracle, bit for bit.
//
// Round 2's router spent 158-175 k cycles per route (profiles/r02_m_route_phase_cycles.txt): a pocket flood and an O(nap^2) pin
// census per route (part of 23 k set-up), a two-hops-per-round-trip back-trace that re-derives every predecessor (39 k), and the
// rounds themselves.  This form keeps round 2's rounds (one barrier per round, mask scan, quads of lanes following chains) — on
// heavy routes nothing tried this round beat them, see DESIGN.md §5.3 — and removes the rest:
//
//  * The field word carries the predecessor direction:  dist << 5 | pdir << 2 | held << 1 | valid.  Lowering is a min() on the
//    WORD, so among candidates of equal distance the lowest pdir (E,S,W,N,U,D = 0..5, the spec's back-trace order,
//    reference baseline/build_3Dgrid.py:127) wins: when the search stops every node with d + h <= best holds its exact distance AND
//    its first tight predecessor (every tight predecessor u of such a node has d(u) + h(u) <= best too, so it was expanded with
//    its final distance and its candidate word took part in the min).  A node is re-opened only when its DISTANCE went down.
//    The back-trace is a pointer chase: one LDS read per node, by one wave, while the others wait at the barrier.
//    27 distance bits: every distance the spec knows is below XR_DIST_CAP = 0x07F00000 ("a distance >= the cap does
//    not exist"), a candidate at or above the cap is never written, and the host only selects this form when one edge plus every
//    penalty stays below 2^20 and the region spans less than 2^25 DBU — so no sum can wrap the word (else: xr_dial.h).
//  * Isolated pins (closed pockets) and the per-net constants (first pin, number of pins) are static: decided at load
//    (xr_batch_load_regions), one word per net — no flood, no pin census per route.
//  * Path nodes are listed while the chase runs and become sources / get claimed in parallel afterwards (no claim mask).
//
//  * XR-Maze v2 (round 4): bit 0 of the word is "outside the net's guide" instead of "valid" (a v2 source word carries bit 3, so only
//    blockages are 0) — the guide test of a hop and of a trace step is a bit of a word that is loaded anyway, decided once per route
//    by one LDS pass over the nodes.  Claims are DEFERRED: an attempt writes nothing to global memory but its path list; whether a
//    node can be claimed is in its word (not held <=> owner 0 or this net), a ripped-up attempt is undone by an LDS-only pass that
//    returns every word to "unreached" (the static bits stay), and the attempt that stands is accepted by one scan of the words
//    (source + owner bit + not held -> owner = net).  No owner reads, no N-wide global passes, no field rebuild per attempt.
//
// Two other search organisations were built on this word format and measured this round (git history of this file): ONE searching
// wave with explicit bucket queues and wave-uniform counters (mean route -30 %, but a route with a wide frontier — 60 rounds — took
// 2.5 M cycles against 1.0 M: 16 quads and list thrash), and a workgroup-wide list form with shared LDS counters (every append /
// pop is a returning LDS atomic inside the hop loop: slower on every route).  Both were bit-exact; neither beat mask rounds where
// it matters: a launch is as long as its longest route.
#pragma once

// XR3_TMP and XR3_LDS_BYTES: xr_device.h (the host sizes the launch from them)
#define XR3_UNREACHED 0xFFFFFFFDu      // | held << 1   (dist bits all ones, pdir 7)
#define XR3_UNREACHED_V2 0xFFFFFFFCu   // | held << 1 | outside the guide   (XR-Maze v2: bit 0 is the guide bit)
#define XR3_DMAX 0x07FFFFFFu           // distance of an unreached word
#define XR3_CAP5 ((uint32_t)(XR_W_USABLE_END >> 2) << 5)     // XR_DIST_CAP x 32: candidates at or above it do not exist (spec)
#define XR3_WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                         __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// Solo rounds (round 5): a round whose predecessor took at most XR3_SOLO_ENTER nodes from the mask is run by ONE wave (the tracing wave) on its
// own — scan, quads, next minimum — and so are the rounds after it, with wave-level synchronisation only, until a round takes more than
// XR3_SOLO_EXIT nodes or the search ends; the other waves of the workgroup are parked at one barrier meanwhile.  A narrow frontier has work for
// a few quads: four waves then mostly pay for each other's barrier skew (26 % of a route).  Same results (rounds are rounds, whoever runs them).
// XR3_SOLO_ENTER < 0: off.
#ifndef XR3_SOLO_ENTER
#define XR3_SOLO_ENTER (-1)
#endif
#ifndef XR3_SOLO_EXIT
#define XR3_SOLO_EXIT 24
#endif

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

// n / d and n % d by the region's exact 24-bit magic (XrRegionDev::m24_*: verified at load for every n asked here): two full-rate
// 24-bit multiplies instead of a 32-bit mul_hi / mul_lo pair plus fix-up
__device__ __forceinline__ void xr3_divmod(uint32_t n, uint32_t d, uint32_t m24, uint32_t sh, uint32_t& q, uint32_t& r) {
    q = (uint32_t)__umul24(n, m24) >> sh;          // (the intrinsic's return type is signed here: shift the unsigned value)
    r = n - (uint32_t)__umul24(q, d);
}

__device__ __forceinline__ int xr3_mbcnt(unsigned long long m) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// V2: XR-Maze v2 knobs compiled in (guide cost, rip-up-and-reroute), as in xr_dial.h
// WIN (round 4, regions too large for LDS — BASELINE config 5): the SAME router inside a window of the region.  The window
// (b.win_x x b.win_y tracks, every layer; fixed per batch so that its index arithmetic is constant) is centred on the box of the net's
// access points; its nodes are loaded with coalesced 16-byte loads, the search, the trace and the claims run on window-local indices, and
// after every search an EXACTNESS CERTIFICATE is checked: face = the smallest key d(u) + len(u, v) + h(v) over the window's boundary
// nodes u and their neighbours v outside (penalty of v unknown: 0, a lower bound).  If face > the distance of the chosen target (or
// nothing was reached and no boundary node was either), no shortest path to any node with d + h <= best leaves the window: take the
// first edge (u, v) out of the window on such a path — its prefix lies inside, so u holds its exact distance here, and by consistency
// of h key(v) <= best, i.e. face <= best.  Then the window's field equals the region's wherever the target choice and the trace look:
// same target, same path, same metrics as the HBM-scratch form / the oracle.  Otherwise (or when the net's box does not fit) the
// function returns false having changed nothing that counts (claims are deferred, the record is written last) and the caller routes
// the net with xr_dial_route_env_big.  Returns true when the step is complete.
template <bool V2, bool WIN = false>
__device__ __forceinline__ bool xr_dial3_route_env(const XrBatchDev& b, const int e, const int a, char* smem) {
    static_assert(!(V2 && WIN), "the window form carries XR-Maze v1 only");
    constexpr bool DEFER = V2 || WIN;                           // claims are written by the attempt that stands, not by the trace
    __shared__ unsigned short s_ap_f[XR_MAX_AP_PER_NET];
    __shared__ short s_ap_pin[XR_MAX_AP_PER_NET];
    __shared__ unsigned char s_ap_conn[XR_MAX_AP_PER_NET];      // 0 target, 1 connected, 2 isolated (static, from the load)
    __shared__ unsigned char s_ap_own[XR_MAX_AP_PER_NET];       // the access point's node had an owner when the route began (a used access point of this net)
    __shared__ unsigned char s_ap_slot[XR_MAX_AP_PER_NET];      // which heuristic box the access point's pin is dealt to (static, from the load)
    __shared__ uint32_t s_min[3], s_bst[3];                     // rotating per round: smallest open key, smallest target distance
    __shared__ int s_cnt[3];                                    // ... and the nodes the round took from the mask (XR3_ADAPT: sparse rounds widen the bucket)
    __shared__ int s_hb[3][6];                                  // heuristic: three boxes over the unconnected pins (x, y: coordinates x32; z), see the search start
    __shared__ int s_qcnt[16];                                  // per wave: nodes of the bucket queued for its quads
    __shared__ int s_hand[2];                                   // solo rounds: where the solo wave left the search (cur | rounds so far << 2), two slots taking turns
    __shared__ unsigned short s_qn[XR_QUAD_POOL];
    __shared__ int s_remaining, s_abort;
    __shared__ int s_gb[4], s_retry, s_ngb;                     // XR-Maze v2: bounding box of the net's access points (track indices), rip-up decision
    __shared__ int4 s_gbx[XR_GUIDE_MAX_BOXES];                  // ... and its guide (xr_guide_load, xr_dial.h): only read by the marking pass
    __shared__ uint32_t s_face;                                 // window form: smallest key of an edge that leaves the window (this search)
    __shared__ int s_fallback;

    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    if (!xr_step_prologue(b, e, a)) return true;

    XR_T0();
    const XrRegionDev R = b.regions[b.env_region[e]];
    const int RX = R.X, RY = R.Y, Z = R.Z;                      // the region; X, Y, N below: the grid the router works on (the window)
    const int RYZ = RY * Z;
    int wx0 = 0, wy0 = 0;                                       // window origin (track indices of the region)
    if (WIN) {
        // ---- does the net fit?  Box of its access points (+ margin) against the window; the window is centred on it, kept inside the
        // region and moved down to a row whose nodes start on a 16-byte boundary of the state rows
        if (tid == 0) { s_gb[0] = 0x7FFFFFFF; s_gb[1] = -1; s_gb[2] = 0x7FFFFFFF; s_gb[3] = -1; s_fallback = 0; }
        __syncthreads();
        const int lo_ = b.net_csr[R.net_off + a], hi_ = b.net_csr[R.net_off + a + 1];
        for (int i = lo_ + tid; i < hi_; i += nthr) {
            const int apf = b.ap_node[R.ap_off + i];
            const int gy = (apf / Z) % RY, gx = apf / RYZ;
            atomicMin(&s_gb[0], gx); atomicMax(&s_gb[1], gx); atomicMin(&s_gb[2], gy); atomicMax(&s_gb[3], gy);
        }
        __syncthreads();
        const int bx0 = s_gb[0], bx1 = s_gb[1], by0 = s_gb[2], by1 = s_gb[3];
        const int spx = bx1 - bx0 + 1, spy = by1 - by0 + 1;
        if (b.win_x > RX || b.win_y > RY || spx + 2 * b.win_margin > b.win_x || spy + 2 * b.win_margin > b.win_y) return false;    // uniform
        wx0 = min(max(bx0 - ((b.win_x - spx) >> 1), 0), RX - b.win_x);
        wy0 = min(max(by0 - ((b.win_y - spy) >> 1), 0), RY - b.win_y) & ~(b.win_ystep - 1);
        if (by1 > wy0 + b.win_y - 1 || by0 < wy0) return false;
        __syncthreads();                                        // (s_gb is initialised again below)
    }
    const int X = WIN ? b.win_x : RX, Y = WIN ? b.win_y : RY, N = WIN ? b.win_x * b.win_y * Z : R.N;
    const int YZ = Y * Z;
    const uint32_t ldir = R.ldir_mask;
    const int mw = (N + 31) >> 5;
    const uint32_t umw = (uint32_t)mw;
    const uint32_t uYZ = (uint32_t)YZ, uZ = (uint32_t)Z;
    const int round_cap = b.round_cap > 0 ? b.round_cap : 1024 + N;
    const int16_t* __restrict__ node_net = b.rg_node_net + R.node_off;
    int16_t* __restrict__ owner = b.owner + (int64_t)e * b.n_max;
    const uint32_t m24_yz = WIN ? b.win_m24_yz : R.m24_yz, m24_mw = WIN ? b.win_m24_mw : R.m24_mw, m24_z = WIN ? b.win_m24_z : R.m24_z;
    const uint32_t s24w = WIN ? b.win_s24 : R.s24;
    // working flat index -> flat index of the region (the state rows, the recorded path, the hash chain)
    auto gflat = [&](int f) __attribute__((always_inline)) -> int {
        if (!WIN) return f;
        uint32_t xw, r;
        xr3_divmod((uint32_t)f, uYZ, m24_yz, s24w & 31u, xw, r);
        return ((wx0 + (int)xw) * RY + wy0) * Z + (int)r;
    };

    // LDS carve:  field u32[n_max] | open | defer | wmin (u32[mw_max] each) | tab u32[x_max+2 + y_max+2] | tmp u16[TMP]
    //   open    bit f: node lowered but not expanded yet (transposed bit order, xr_dial.h: node f <-> word f % mw, bit f / mw)
    //   defer   bit f: an edge out of f was refused only because of the search bound; re-opened when the next search starts
    //   wmin[w] lower bound of the keys of the open nodes of word w (XR_DIAL_INF: none)
    uint32_t* field = reinterpret_cast<uint32_t*>(smem);
    const int n_work = WIN ? b.win_nmax : b.n_max;              // (window form: the carve of XR3_LDS_BYTES(win_nmax, win_x, win_y))
    const int mw_max = (n_work >> 5) + 1;
    uint32_t* s_open = field + n_work;
    uint32_t* s_defer = s_open + mw_max;
    uint32_t* s_wmin = s_defer + mw_max;
    // coordinate tables (x32, relative to the first track): tab[k] = 32*(xs[clamp(k-1)] - xs[0]), k = 0 .. X+1 (the coordinate of
    // track x is tab[x+1]; padded at both ends); the y table follows at XO
    uint32_t* s_tab = s_wmin + mw_max;
    const int XO = (WIN ? b.win_x : b.x_max) + 2;
    unsigned short* s_tmp = reinterpret_cast<unsigned short*>(s_tab + XO + (WIN ? b.win_y : b.y_max) + 2);

    // loads that depend on (e, a) only: issued now, consumed after the grid build
    const int ap_lo = b.net_csr[R.net_off + a], ap_hi = b.net_csr[R.net_off + a + 1];
    const int nap = ap_hi - ap_lo;    // 1 <= nap <= XR_MAX_AP_PER_NET (checked at load)
    const int ninfo = b.net_info[R.net_off + a];               // first pin | pins << 14 | isolated pins << 22 | first pin isolated << 30
    const int meas_cls = (b.heavy_class > 0 && b.net_meas) ? (int)b.net_meas[R.net_off + a] : 0;      // what this net cost the last time it was routed
    int my_ap_f = 0, my_ap_pin = 0, my_ap_iso = 0;
    if (tid < nap) {
        my_ap_f = b.ap_node[R.ap_off + ap_lo + tid]; my_ap_pin = b.ap_pin[R.ap_off + ap_lo + tid];
        my_ap_iso = b.ap_flags[R.ap_off + ap_lo + tid];
    }
    // (window form: entry k is the track wx0 + k - 1 of the REGION — the two pad entries hold the coordinates of the tracks just outside
    //  the window where the region goes on: what the certificate's edge lengths need — relative to the lowest of them)
    const int cbx = b.coords[R.xs_off + max(wx0 - 1, 0)], cby = b.coords[R.ys_off + max(wy0 - 1, 0)];
    uint32_t my_xc = 0, my_yc = 0;
    if (tid <= X + 1) my_xc = (uint32_t)(b.coords[R.xs_off + min(max(wx0 + tid - 1, 0), RX - 1)] - cbx) << 5;
    if (tid <= Y + 1) my_yc = (uint32_t)(b.coords[R.ys_off + min(max(wy0 + tid - 1, 0), RY - 1)] - cby) << 5;
    const int first_pin = ninfo & 0x3FFF, npins = (ninfo >> 14) & 0xFF;
    const int n_isolated = (ninfo >> 30) & 1 ? npins - 1 : (ninfo >> 22) & 0xFF;      // unreachable pins known up front

    // XR-Maze v2, guide membership (round 5): a static bit per node of this (region, net) — "outside the guide" — read with the grid build:
    // one byte per chunk of 8 nodes, in flight with the chunk's state loads (round 4: the net's boxes fetched per route by two dependent
    // global loads and tested in a pass of their own).  Null: no static masks (xr_batch.cpp build_guide_masks) -> the pass below.
    const uint8_t* __restrict__ gmrow = (V2 && b.guide_cost && b.guide_mask) ? b.guide_mask + R.gmask_off + (int64_t)(a - 1) * R.gmask_stride : nullptr;
    // ---- grid build: field word of every node for THIS net (16-byte loads of node_net / owner, four chunks in flight) ----
    auto build_field = [&]() __attribute__((always_inline)) {
        const int nchunk = (N + 7) >> 3;
        for (int c0 = tid; c0 < nchunk; c0 += 4 * nthr) {
            int4 vn[4], vo[4];
            uint32_t gm[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int ci = c0 + u * nthr;
                if (ci < nchunk) {
                    const int g0 = gflat(ci << 3);            // (window form: a chunk of 8 never straddles a window row — (win_y * Z) % 8 == 0 — and
                                                              //  starts on a 16-byte boundary of the rows: wy0 is a multiple of win_ystep)
                    vn[u] = *reinterpret_cast<const int4*>(node_net + g0);
                    vo[u] = *reinterpret_cast<const int4*>(owner + g0);
                    if (V2 && gmrow) gm[u] = gmrow[ci];
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
                    const uint32_t ww = (V2 ? (XR3_UNREACHED_V2 | ((gm[u] >> j) & 1u)) : XR3_UNREACHED) | (((ow != 0 && ow != a) || (nn > 0 && nn != a)) ? 2u : 0u);
                    w[j] = (nn == -1 || f0 + j >= N) ? 0u : ww;
                }
                uint4* dst = reinterpret_cast<uint4*>(field + f0);
                dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
                dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
            }
        }
        for (int i = tid; i < mw; i += nthr) { s_open[i] = 0; s_defer[i] = 0; s_wmin[i] = XR_DIAL_INF; }
    };
    build_field();
    // (consumed after the barrier below; under a saturated write stream every dependent global round trip of a route costs ~1 us,
    //  so the searches themselves never read global memory: what they need to know about `owner` is in the field words)
    int my_ap_own = 0;
    if (tid < nap) my_ap_own = owner[my_ap_f];
    if (tid <= X + 1) s_tab[tid] = my_xc;
    if (tid <= Y + 1) s_tab[XO + tid] = my_yc;
    for (int i = tid + nthr; i <= X + 1; i += nthr)
        s_tab[i] = (uint32_t)(b.coords[R.xs_off + min(wx0 + i - 1, RX - 1)] - cbx) << 5;
    for (int i = tid + nthr; i <= Y + 1; i += nthr)
        s_tab[XO + i] = (uint32_t)(b.coords[R.ys_off + min(wy0 + i - 1, RY - 1)] - cby) << 5;
    if (tid == 0) { s_gb[0] = 0x7FFFFFFF; s_gb[1] = -1; s_gb[2] = 0x7FFFFFFF; s_gb[3] = -1; }
    const bool guide_pass = V2 && b.guide_cost && !gmrow;        // (uniform) membership decided per route: only without the static masks
    if (guide_pass) __syncthreads();
    for (int i = tid; i < nap; i += nthr) {
        const int pin = i < nthr ? my_ap_pin : (int)b.ap_pin[R.ap_off + ap_lo + i];
        const int apf = i < nthr ? my_ap_f : b.ap_node[R.ap_off + ap_lo + i];
        const int iso = i < nthr ? my_ap_iso : (int)b.ap_flags[R.ap_off + ap_lo + i];
        s_ap_f[i] = (unsigned short)(WIN ? ((apf / RYZ - wx0) * Y + ((apf / Z) % RY - wy0)) * Z + apf % Z : apf);
        s_ap_pin[i] = (short)pin;
        s_ap_conn[i] = (unsigned char)((iso & 1) ? 2 : (pin == first_pin ? 1 : 0));
        s_ap_slot[i] = (unsigned char)((iso >> 1) & 3);
        s_ap_own[i] = (unsigned char)((i < nthr ? my_ap_own : (int)owner[apf]) != 0);
        if (guide_pass) {                        // XR-Maze v2: the net's default guide = bounding box of all its access points (+ margin)
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
    const uint32_t delta = R.w_min * (uint32_t)b.dial_mult * (uint32_t)(meas_cls >= b.heavy_class && b.heavy_class > 0 ? b.heavy_mult : 1);  // bucket width (keys f = d + h, DBU)
    const uint32_t guide5 = V2 ? (uint32_t)b.guide_cost << 5 : 0u;
    const uint32_t sh_yz = s24w & 31u, sh_z = (s24w >> 8) & 31u, sh_mw = (s24w >> 16) & 31u;       // (window form: its own magics)
    auto node_xyz = [&](uint32_t f, int& x, int& y, int& z) __attribute__((always_inline)) {
        uint32_t ux, ur, uy, uz;
        xr3_divmod(f, uYZ, m24_yz, sh_yz, ux, ur);
        xr3_divmod(ur, uZ, m24_z, sh_z, uy, uz);
        x = (int)ux; y = (int)uy; z = (int)uz;
    };
    if (guide_pass) {
        // ---- guide membership as data (round 4): ONE pass over the nodes sets bit 0 of every word outside the net's guide (<= 8 box
        // tests per node, boxes unpacked once per 8-node chunk); a hop and a trace step then test a bit of a word they load anyway.
        xr_guide_load(b, R, a, s_gb, Z, s_gbx, &s_ngb, tid);
        __syncthreads();
        const int ngb = s_ngb;
        const int nchunk = (N + 7) >> 3;
        for (int ci = tid; ci < nchunk; ci += nthr) {
            const int f0 = ci << 3;
            int x, y, z;
            node_xyz((uint32_t)f0, x, y, z);
            uint32_t in = 0u;                                  // bit j: node f0 + j lies inside some box
            for (int g = 0; g < ngb; g++) {
                const int4 bx = s_gbx[g];
                const int x0 = bx.x & 0xFFFF, x1 = (int)((uint32_t)bx.x >> 16), y0 = bx.y & 0xFFFF, y1 = (int)((uint32_t)bx.y >> 16);
                const int z0 = bx.z & 0xFFFF, z1 = (int)((uint32_t)bx.z >> 16);
                int xx = x, yy = y, zz = z;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    in |= (uint32_t)(xx >= x0 && xx <= x1 && yy >= y0 && yy <= y1 && zz >= z0 && zz <= z1) << j;
                    if (++zz == Z) { zz = 0; if (++yy == Y) { yy = 0; xx++; } }
                }
            }
            if ((in & 0xFFu) != 0xFFu) {
                uint4* dst = reinterpret_cast<uint4*>(field + f0);
                uint4 lo = dst[0], hi = dst[1];
                lo.x |= (lo.x != 0u && !(in & 1u)) ? 1u : 0u;   lo.y |= (lo.y != 0u && !(in & 2u)) ? 1u : 0u;
                lo.z |= (lo.z != 0u && !(in & 4u)) ? 1u : 0u;   lo.w |= (lo.w != 0u && !(in & 8u)) ? 1u : 0u;
                hi.x |= (hi.x != 0u && !(in & 16u)) ? 1u : 0u;  hi.y |= (hi.y != 0u && !(in & 32u)) ? 1u : 0u;
                hi.z |= (hi.z != 0u && !(in & 64u)) ? 1u : 0u;  hi.w |= (hi.w != 0u && !(in & 128u)) ? 1u : 0u;
                dst[0] = lo; dst[1] = hi;
            }
        }
        __syncthreads();
    }
    int32_t* __restrict__ path = b.path + (int64_t)e * b.path_cap;
    // XR-Maze v2's rip-up-and-reroute loop has ONE possible outcome: the LAST attempt's route (round 5; DESIGN.md §3.1 "the loop is one
    // attempt").  If attempt t stands because its paths use no held node, attempt t + 1 would repeat it search by search: paths without a
    // held node cost what they cost, every alternative through one only got dearer — same distances along them, same first tight
    // predecessors, same targets (the argument behind round 4's "resume at the failed search", applied to a whole attempt) — and so would
    // every later one, up to the last.  If no attempt stands, the last one is the result by definition.  So the route is computed once, at
    // the penalty of the last attempt (pen << (maze_end_iter - 1)); the loop below never iterates.  -DXR3_V2_ALL_ATTEMPTS keeps the
    // attempt-by-attempt form of round 4 (same results: the A/B and the proof by test).
#ifdef XR3_V2_ALL_ATTEMPTS
    int attempt = 0;
#else
    int attempt = V2 ? max(b.maze_end_iter, 1) - 1 : 0;
    if (V2) pen5 <<= attempt;
#endif

    // per-lane constants of a quad: lane 4g + d relaxes direction d of quad g's node: 0 +planar, 1 -planar, 2 +z, 3 -z
    const int dir = lane & 3, qbase = lane & ~3;
    const int sgn = (dir & 1) ? -1 : 1;
    const bool planar = dir < 2;
    const int stepH = planar ? sgn * YZ : sgn, stepV = planar ? sgn * Z : sgn;           // flat-index step on a horizontal / vertical layer
    const int limH = planar ? X : Z, limV = planar ? Y : Z;
    // predecessor direction stored in the NEIGHBOUR's word (pointing back at the quad's node), E,S,W,N,U,D = 0..5
    const uint32_t pdH = (planar ? (sgn > 0 ? 2u : 0u) : (sgn > 0 ? 5u : 4u)) << 2, pdV = (planar ? (sgn > 0 ? 1u : 3u) : (sgn > 0 ? 5u : 4u)) << 2;

    // node f <-> (word f % mw, bit f / mw) of the node bitmasks
    auto mask_pos = [&](uint32_t f, uint32_t& q, uint32_t& r) __attribute__((always_inline)) { xr3_divmod(f, umw, m24_mw, sh_mw, q, r); };
    // a node becomes a source: distance 0, open
    // a node becomes a source: distance 0.  Bits 2..4 of a source word carry no predecessor; bit 2 = "the node has an owner" (it was
    // claimed by this route, or was held before it began): the terminal node of a trace is claimed iff it has none, and the answer must
    // not cost a global load.  A node that already is a source keeps its bit.
    auto make_source = [&](uint32_t f, bool owned) __attribute__((always_inline)) {
        uint32_t q, r;
        mask_pos(f, q, r);
        const uint32_t w0 = field[f];
        field[f] = (w0 & ((w0 >> 5) == 0u ? 7u : 3u)) | (owned ? 4u : 0u) | (V2 ? 8u : 0u);    // (v2: bit 0 is the guide bit, so bit 3 keeps a source != 0)
        atomicOr(&s_open[r], 1u << q);
        s_wmin[r] = 0u;                       // (racing plain stores of the same value)
    };

    const uint64_t h0 = wv == sw ? b.hash[e] : 0ULL;        // (a ripped-up attempt rewinds the chain to here)
    // the tracing wave's bookkeeping (uniform over that wave; meaningless in the others), and its state at the start of the search that
    // is running (XR-Maze v2: where a ripped-up attempt resumes, see the rip-up below)
    int d_vio = 0, d_wl = 0, d_via = 0, plen = 0, status = XR_ENV_OK, d_held = 0, nrounds = 0;
    uint64_t h = h0;
    int sv_vio = 0, sv_wl = 0, sv_via = 0, sv_plen = 0, sv_status = XR_ENV_OK, sv_rem = 0;
    uint64_t sv_h = h0;
    bool resume = false;
    bool win_fail = false;                                  // window form, tracing wave: a certificate failed
    for (;;) {                                              // attempts (exactly one unless maze_end_iter > 1)
    // component = all access points of the lowest pin id (a resumed attempt: of every pin connected so far; its path nodes are sources still)
    for (int i = tid; i < nap; i += nthr)
        if (s_ap_conn[i] == 1) make_source((uint32_t)s_ap_f[i], s_ap_own[i] != 0);
    if (tid == 0) { s_remaining = resume ? sv_rem : npins - 1 - n_isolated; s_abort = 0; }
    if (resume) { d_vio = sv_vio; d_wl = sv_wl; d_via = sv_via; plen = sv_plen; status = sv_status; h = sv_h; }
    else { d_vio = 0; d_wl = 0; d_via = 0; plen = 0; status = XR_ENV_OK; h = h0; }
    d_held = 0; nrounds = 0;

    for (;;) {
        // ---- new search: sources are open with distance 0; deferred nodes are looked at again ------------
        if (tid == 0) {
            s_min[0] = 0; s_min[1] = XR_DIAL_INF; s_min[2] = XR_DIAL_INF;
            s_bst[0] = XR_DIAL_INF; s_bst[1] = XR_DIAL_INF; s_bst[2] = XR_DIAL_INF;
            s_cnt[0] = 1 << 20; s_cnt[1] = 0; s_cnt[2] = 0;
        }
        if (tid < 18) { const int k = tid % 6; s_hb[tid / 6][k] = (k & 1) ? -0x3FFFFFFF : 0x3FFFFFFF; }      // (empty: min > max)
        for (int i = tid; i < mw; i += nthr) {
            const uint32_t m = s_defer[i];
            if (m) { atomicOr(&s_open[i], m); s_defer[i] = 0; s_wmin[i] = 0u; }   // (0: a lower bound; the first scan fixes it)
        }
        xr_lds_barrier();
        if (s_remaining <= 0) break;          // uniform: written before the barrier above
        if (V2) { sv_vio = d_vio; sv_wl = d_wl; sv_via = d_via; sv_plen = plen; sv_status = status; sv_h = h; sv_rem = s_remaining; }
        // heuristic of this search (round 4): the unconnected (and not isolated) pins are dealt over THREE boxes (statically, by pin
        // rank: ap_flags) and h(v) = the smallest distance to a box — coordinate differences + one via cost per layer, from coordinates
        // x32.  A minimum of consistent lower bounds is one, and it is 0 on every target: same results as with one box around all of
        // them — but one box is 0 everywhere between the pins of a spread multi-pin net, which then floods its whole box.
        for (int i = tid; i < nap; i += nthr)
            if (!s_ap_conn[i]) {
                int ax, ay, az;
                node_xyz((uint32_t)s_ap_f[i], ax, ay, az);
                const int cx = (int)s_tab[ax + 1], cy = (int)s_tab[XO + ay + 1];
#ifdef XR3_HB_ONE       // A/B: one box around all unconnected pins (rounds 2-3)
                int* hb = s_hb[0];
#else
                int* hb = s_hb[s_ap_slot[i] < 3 ? s_ap_slot[i] : 0];
#endif
                atomicMin(&hb[0], cx); atomicMax(&hb[1], cx); atomicMin(&hb[2], cy); atomicMax(&hb[3], cy);
                atomicMin(&hb[4], az); atomicMax(&hb[5], az);
            }
        xr_lds_barrier();
        int hbv[3][6];
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int j = 0; j < 6; j++) hbv[k][j] = __builtin_amdgcn_readfirstlane(s_hb[k][j]);       // (uniform: scalar registers)
        const bool hb0on = hbv[0][0] <= hbv[0][1], hb1on = hbv[1][0] <= hbv[1][1], hb2on = hbv[2][0] <= hbv[2][1];      // (an empty box is skipped)
        auto heur_1 = [&](const int* hb, int xc, int yc, int z) __attribute__((always_inline)) -> uint32_t {
            const int hx = max(0, max(hb[0] - xc, xc - hb[1])), hy = max(0, max(hb[2] - yc, yc - hb[3]));
            const int hz = max(0, max(hb[4] - z, z - hb[5]));
            return (((uint32_t)hx + (uint32_t)hy) >> 5) + __umul24((uint32_t)hz, (uint32_t)b.via_cost);
        };
        auto heur_c = [&](int xc, int yc, int z) __attribute__((always_inline)) -> uint32_t {
            uint32_t hv = 0xFFFFFFFFu;
            if (hb0on) hv = heur_1(hbv[0], xc, yc, z);
            if (hb1on) { const uint32_t t = heur_1(hbv[1], xc, yc, z); hv = t < hv ? t : hv; }
            if (hb2on) { const uint32_t t = heur_1(hbv[2], xc, yc, z); hv = t < hv ? t : hv; }
            return hv;
        };
        XR_LAP(4);
        int cur = 0;
        [[maybe_unused]] int bscale = 1;      // (XR3_ADAPT_LO builds: bucket width of the round, in units of delta)
        bool aborted = false;                 // (round cap: xr_dial.h)
        bool soloing = false;                 // this wave runs the rounds on its own (the tracing wave only; the others wait at a barrier)
        int hpar = 0;                         // which hand-off slot the next solo episode uses (every wave counts the episodes)
        for (int nsr = 0;; nsr++) {
            int nx1 = cur == 2 ? 0 : cur + 1, nx2 = nx1 == 2 ? 0 : nx1 + 1;
            const uint32_t m = s_min[cur], best = s_bst[cur];
            const bool sdone = m == XR_DIAL_INF || m > best;
            const bool scap = nsr >= round_cap || s_abort;
            if (XR3_SOLO_ENTER >= 0) {
                const int pc = s_cnt[cur];                        // nodes the round before this one took from the mask (search start: 1 << 20)
                if (soloing && (sdone || scap || pc > XR3_SOLO_EXIT)) {      // leave the episode: publish where the search stands, release the others
                    if (lane == 0) s_hand[hpar] = cur | (nsr << 2);
                    hpar ^= 1;
                    xr_lds_barrier();
                    soloing = false;
                }
                if (!soloing && !sdone && !scap && pc <= XR3_SOLO_ENTER) {   // (uniform over the workgroup: everybody read the same pc)
                    if (wv != sw) {                               // parked until the solo wave leaves its episode
                        xr_lds_barrier();
                        const int hv = s_hand[hpar];
                        hpar ^= 1;
                        cur = hv & 3; nsr = (hv >> 2) - 1;
                        continue;
                    }
                    soloing = true;
                }
            }
            if (sdone) break;                                                // uniform
            if (scap) { aborted = true; break; }                             // uniform (s_abort: written before the last barrier)
            // the team of this round: the whole workgroup, or the solo wave
            const int tE = soloing ? lane : tid, nE = soloing ? 64 : nthr;
#ifdef XR3_GROW_AFTER        // A/B only (profiles/r03_q_ab_growing_bucket_width.txt: wider late buckets cost 3-18 % --
                             // the extra re-expansions outweigh the rounds saved); off in the shipped build
            const uint32_t hi = m + (nsr >= XR3_GROW_AFTER ? delta * XR3_GROW_BY : delta);
#elif defined(XR3_ADAPT_LO)  // A/B (round 4): a round that took fewer than XR3_ADAPT_LO nodes from the mask doubles the next bucket (up to
                             // XR3_ADAPT_MAX x), one that took more than XR3_ADAPT_HI goes back to one width.  Bucket widths never change results.
            { const int pc = s_cnt[cur]; if (pc < XR3_ADAPT_LO) bscale = min(bscale << 1, XR3_ADAPT_MAX); else if (pc > XR3_ADAPT_HI) bscale = 1; }
            const uint32_t hi = m + delta * (uint32_t)bscale;
#else
            const uint32_t hi = m + delta;
#endif
            uint32_t lmin = XR_DIAL_INF;
            if (tE == 0) { s_min[nx2] = XR_DIAL_INF; s_bst[nx2] = XR_DIAL_INF; s_cnt[nx2] = 0; }
            // bound for the next round: smallest tentative distance of an unconnected target
            // (by the threads at the END of the workgroup: the first wave carries the words beyond one per thread)
            for (int i = nE - 1 - tE; i < nap; i += nE)
                if (!s_ap_conn[i]) { const uint32_t d = field[s_ap_f[i]] >> 5; if (d != XR3_DMAX) atomicMin(&s_bst[nx1], d); }
            // A lane scans word wi and — where the mask has more words than the workgroup has threads — word wi + nthr in the
            // SAME pass (one 64-bit bit set).  The nodes of this bucket are then expanded by QUADS of lanes, one lane per direction.
            for (int wbase = 0; wbase < mw; wbase += 2 * nE) {            // (uniform trip count: the expansion is wave-cooperative)
                const int wi = wbase + tE;
                const int wi2 = wi + nE;
                unsigned long long expd = 0;
                if (wi < mw) {
                    const bool has2 = wi2 < mw;
                    uint32_t wmA = s_wmin[wi], wmB = has2 ? s_wmin[wi2] : XR_DIAL_INF;
                    const bool actA = wmA < hi, actB = wmB < hi;
                    if (actA || actB) {
                        // the word (probably) holds a node of this bucket: take it.  Order matters: reset the cached minimum,
                        // THEN take the bits, THEN read distances — a concurrent insertion is either seen here or survives
                        uint32_t bA = 0, bB = 0;
                        if (actA) { s_wmin[wi] = XR_DIAL_INF; bA = atomicExch(&s_open[wi], 0u); }
                        if (actB) { s_wmin[wi2] = XR_DIAL_INF; bB = atomicExch(&s_open[wi2], 0u); }
                        unsigned long long bits = (unsigned long long)bA | ((unsigned long long)bB << 32);
                        unsigned long long keep = 0;
                        uint32_t kminA = XR_DIAL_INF, kminB = XR_DIAL_INF;
                        while (bits) {                                    // XR_SCAN_UNROLL distance loads in flight at a time
                            int q[XR_SCAN_UNROLL];
                            uint32_t w[XR_SCAN_UNROLL];
#pragma unroll
                            for (int j = 0; j < XR_SCAN_UNROLL; j++) {
                                q[j] = bits ? __ffsll((long long)bits) - 1 : -1;
                                bits &= bits - 1;                        // (0 stays 0)
                            }
                            int fq[XR_SCAN_UNROLL];
                            int cx[XR_SCAN_UNROLL], cy[XR_SCAN_UNROLL], cz[XR_SCAN_UNROLL];
                            uint32_t xcq[XR_SCAN_UNROLL], ycq[XR_SCAN_UNROLL];
#pragma unroll
                            for (int j = 0; j < XR_SCAN_UNROLL; j++) fq[j] = q[j] >= 0 ? (q[j] & 31) * mw + (q[j] < 32 ? wi : wi2) : 0;
#pragma unroll
                            for (int j = 0; j < XR_SCAN_UNROLL; j++) w[j] = field[fq[j]];
#pragma unroll
                            for (int j = 0; j < XR_SCAN_UNROLL; j++) node_xyz((uint32_t)fq[j], cx[j], cy[j], cz[j]);
#pragma unroll
                            for (int j = 0; j < XR_SCAN_UNROLL; j++) { xcq[j] = s_tab[cx[j] + 1]; ycq[j] = s_tab[XO + cy[j] + 1]; }
#pragma unroll
                            for (int j = 0; j < XR_SCAN_UNROLL; j++) {
                                if (q[j] < 0) continue;
                                const uint32_t key = (w[j] >> 5) + heur_c((int)xcq[j], (int)ycq[j], cz[j]);
                                if (key >= hi) {
                                    keep |= 1ULL << q[j];
                                    if (q[j] < 32) kminA = key < kminA ? key : kminA; else kminB = key < kminB ? key : kminB;
                                } else expd |= 1ULL << q[j];
                            }
                        }
                        if ((uint32_t)keep) { atomicOr(&s_open[wi], (uint32_t)keep); atomicMin(&s_wmin[wi], kminA); }
                        if ((uint32_t)(keep >> 32)) { atomicOr(&s_open[wi2], (uint32_t)(keep >> 32)); atomicMin(&s_wmin[wi2], kminB); }
                        if (actA) wmA = kminA;
                        if (actB) wmB = kminB;
                    }
                    lmin = wmA < lmin ? wmA : lmin;
                    lmin = wmB < lmin ? wmB : lmin;
                }
                XR_LAP(1);
                // ---- the wave's nodes of this bucket go into its slice of a small LDS queue (no room: back into the mask)
                {
                    const int qcap = soloing ? XR_QUAD_POOL : XR_QUAD_POOL / ((nthr + 63) >> 6);      // (alone: the whole pool)
                    int* qcnt = &s_qcnt[wv];
                    unsigned short* qn = soloing ? s_qn : s_qn + wv * qcap;
                    if (lane == 0) *qcnt = 0;
                    __builtin_amdgcn_wave_barrier();
                    while (expd) {
                        const int qb = __ffsll((long long)expd) - 1;
                        expd &= expd - 1;
                        const int wsel = qb < 32 ? wi : wi2;
                        const uint32_t f = (uint32_t)((qb & 31) * mw + wsel);
                        const int pos = atomicAdd(qcnt, 1);
                        if (pos < qcap) qn[pos] = (unsigned short)f;
                        else { atomicOr(&s_open[wsel], 1u << (qb & 31)); atomicMin(&s_wmin[wsel], m); lmin = m < lmin ? m : lmin; }
                    }
                    __builtin_amdgcn_wave_barrier();
                    const int nq = min(__builtin_amdgcn_readfirstlane(*qcnt), qcap);
#if defined(XR3_ADAPT_LO)
                    if (lane == 0 && nq) atomicAdd(&s_cnt[nx1], nq);
#else
                    if (XR3_SOLO_ENTER >= 0 && lane == 0 && nq) atomicAdd(&s_cnt[nx1], nq);
#endif
                    // ---- quads: lanes 4g .. 4g+3 follow ONE chain, lane 4g+d relaxes direction d of the chain's current node
                    int gf = -1, gx = 0, gy = 0, gz = 0, qh = 0;
                    for (int nhop = 0;; nhop++) {
                        if (nhop >= round_cap) {          // (every hop lowers a field word: finite anyway; the cap is the hang guard)
                            if (gf >= 0 && dir == 0) { uint32_t oq, orr; mask_pos((uint32_t)gf, oq, orr); atomicOr(&s_open[orr], 1u << oq); s_abort = 1; }
                            break;
                        }
                        const unsigned long long idle_g = __ballot(gf < 0) & 0x1111111111111111ULL;       // one bit per idle quad
                        if (gf < 0) {
                            const int idx = qh + (int)__popcll(idle_g & ((1ULL << qbase) - 1ULL));
                            if (idx < nq) { gf = (int)qn[idx]; node_xyz((uint32_t)gf, gx, gy, gz); }
                        }
                        qh += (int)__popcll(idle_g);
                        if (__ballot(gf >= 0) == 0ULL) break;                                            // uniform
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
                        const int nz = gz + (planar ? 0 : sgn);
                        const uint32_t cand5 = (gw & ~31u) + len5 + ((wn & 2u) ? pen5 : 0u) + ((V2 && (wn & 1u)) ? guide5 : 0u);
                        const uint32_t cw = cand5 | (vert ? pdV : pdH) | (wn & 3u);
                        const uint32_t key = (cand5 >> 5) + heur_c((int)((planar && !vert) ? cb : xq), (int)((planar && vert) ? cb : yq), nz);   // f = d + h
                        // blockage, or no improvement of the WORD (distance, then predecessor direction): nothing to do
                        const bool go = inb && wn != 0u && gw < 0xFFFFFFE0u && cand5 < XR3_CAP5 && cw < wn;     // (gw reached => below the cap: the sum cannot wrap)
                        const bool refused = go && key > best;                                 // bound pruning (on f)
                        bool improved = false;
                        if (go && !refused) {
                            const uint32_t old = atomicMin(&field[nf], cw);
                            improved = (cw >> 5) < (old >> 5);                                 // the DISTANCE went down (not only the direction)
                        }
                        const bool chain_cand = improved && key < hi;
                        const uint32_t c4 = (uint32_t)(__ballot(chain_cand) >> qbase) & 15u;
                        const uint32_t r4 = (uint32_t)(__ballot(refused) >> qbase) & 15u;
                        const int win = c4 ? __ffs((int)c4) - 1 : -1;                          // the chain goes on with the first lowered direction
                        if (improved && dir != win) {                                          // the others become open
                            uint32_t oq, orr;
                            mask_pos((uint32_t)nf, oq, orr);
                            atomicOr(&s_open[orr], 1u << oq);
                            atomicMin(&s_wmin[orr], key);
                            lmin = key < lmin ? key : lmin;
                        }
                        if (act) {
                            if (r4 && dir == 0) { uint32_t oq, orr; mask_pos((uint32_t)gf, oq, orr); atomicOr(&s_defer[orr], 1u << oq); }
                            if (win >= 0) {                                 // every lane of the quad moves to the winner's node
                                const int sw_ = (win & 1) ? -1 : 1;
                                const bool wpl = win < 2;
                                const int stp = wpl ? (vert ? Z : YZ) : 1;             // (no multiplies: the step is one of three constants)
                                gf += sw_ > 0 ? stp : -stp;
                                gx += (wpl && !vert) ? sw_ : 0; gy += (wpl && vert) ? sw_ : 0; gz += wpl ? 0 : sw_;
                            } else gf = -1;
                        }
                    }
                }
                XR_LAP(2);
            }
            lmin = xr3_wave_min(lmin);
            if (lane == 0 && lmin != XR_DIAL_INF) atomicMin(&s_min[nx1], lmin);
            if (wv == sw) nrounds++;
            if (soloing) XR3_WSYNC(); else xr_lds_barrier();
            XR_LAP(6);
            cur = nx1;
        }

        if (WIN) {
            // ---- exactness certificate of this search (see the function's header): the smallest key of an edge that leaves the window.
            // Every boundary node the search reached, with its distance as it stands (a node with d + h <= best holds its exact one), the
            // true length of the edge to the track outside (pad entries of the coordinate tables), no penalty for the node out there.
            if (tid == 0) s_face = XR_DIAL_INF;
            xr_lds_barrier();
            uint32_t fm = XR_DIAL_INF;
            const int XZ = X * Z;
            for (int i = tid; i < 2 * YZ; i += nthr) {                  // x faces (columns 0 and X - 1): edges of horizontal layers
                const int side = i >= YZ ? 1 : 0, r = side ? i - YZ : i;
                if (!(side ? wx0 + X < RX : wx0 > 0)) continue;         // the region ends here too
                uint32_t yw, z;
                xr3_divmod((uint32_t)r, uZ, m24_z, sh_z, yw, z);
                if ((ldir >> z) & 1u) continue;
                const uint32_t w = field[(side ? (X - 1) * YZ : 0) + r];
                if (w == 0u || (w >> 5) == XR3_DMAX) continue;
                const uint32_t cin = s_tab[side ? X : 1], cout = s_tab[side ? X + 1 : 0];
                const uint32_t key = (w >> 5) + ((side ? cout - cin : cin - cout) >> 5) + heur_c((int)cout, (int)s_tab[XO + yw + 1], (int)z);
                fm = key < fm ? key : fm;
            }
            for (int i = tid; i < 2 * XZ; i += nthr) {                  // y faces (rows 0 and Y - 1): edges of vertical layers
                const int side = i >= XZ ? 1 : 0, r = side ? i - XZ : i;
                if (!(side ? wy0 + Y < RY : wy0 > 0)) continue;
                uint32_t xw, z;
                xr3_divmod((uint32_t)r, uZ, m24_z, sh_z, xw, z);
                if (!((ldir >> z) & 1u)) continue;
                const uint32_t w = field[((int)xw * Y + (side ? Y - 1 : 0)) * Z + (int)z];
                if (w == 0u || (w >> 5) == XR3_DMAX) continue;
                const uint32_t cin = s_tab[XO + (side ? Y : 1)], cout = s_tab[XO + (side ? Y + 1 : 0)];
                const uint32_t key = (w >> 5) + ((side ? cout - cin : cin - cout) >> 5) + heur_c((int)s_tab[xw + 1], (int)cout, (int)z);
                fm = key < fm ? key : fm;
            }
            fm = xr3_wave_min(fm);
            if (lane == 0 && fm != XR_DIAL_INF) atomicMin(&s_face, fm);
            xr_lds_barrier();
        }

        // ===================== the tracing wave: target, back-trace, new sources (the others go on to the next barrier) ==========
        if (wv == sw) {
            int remaining = s_remaining;
            if (WIN && aborted) { win_fail = true; remaining = 0; }          // (a round cap inside a window: let the HBM-scratch form decide)
            else if (aborted) {                   // round cap: the remaining pins are charged as unreachable, nothing is traced
                d_vio += remaining; status |= XR_ENV_ROUTER_ABORT | XR_ENV_UNREACHABLE; remaining = 0;
            } else {
                // ---- nearest access point of an unconnected pin; ties -> lowest flat index --------------------------------
                uint32_t md = XR3_DMAX;
                for (int i = lane; i < nap; i += 64)
                    if (!s_ap_conn[i]) { const uint32_t d = field[s_ap_f[i]] >> 5; md = d < md ? d : md; }
                const uint32_t bd = xr3_wave_min(md);
                if (WIN && s_face <= bd) { win_fail = true; remaining = 0; }   // certificate failed (bd = XR3_DMAX when nothing was reached: any finite face key fails it)
                else if (bd == XR3_DMAX) {        // every remaining pin unreachable
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
                    // XR-Maze v2: an attempt whose path uses a held node is ripped up unless it is the last one — nothing of it is kept,
                    // so it may as well end at the first held node its trace meets (same results, the doomed attempt's remaining
                    // trace and searches are skipped)
                    const bool doomable = V2 && attempt + 1 < b.maze_end_iter;
                    bool doomed = false, dirty = false;        // dirty: part of the doomed trace was already flushed (its nodes are sources)
                    auto flush = [&]() __attribute__((always_inline)) {        // the listed path nodes: sources of the next search, claimed if nobody holds them
                        XR3_WSYNC();
                        const int pl0 = plen - np;              // (the listed nodes are path[pl0 .. plen): one coalesced store per 64 nodes —
                                                                //  a store per hop queues behind the unit writers' stores of the same CU)
                        for (int i = lane; i < np; i += 64) {
                            const uint32_t f = s_tmp[i];
                            const uint32_t w0 = field[f];
                            if (pl0 + i < b.path_cap) path[pl0 + i] = gflat((int)f);
                            make_source(f, true);               // (whatever it was before: it has an owner from here on)
                            // claimed if nobody OWNS it.  Not held (bit 1 clear) = owner 0 or this net itself (a used access point of
                            // it): the store of `a` is right in both cases and needs no load.  Held = owned by another net OR another
                            // net's still unowned access point: only then the owner is read.  XR-Maze v2 defers all of this to the attempt
                            // that stands (the acceptance scan below finds these words: source + owner bit).
                            if (!DEFER) { if (!(w0 & 2u)) owner[f] = (int16_t)a; else if (owner[f] == 0) owner[f] = (int16_t)a; }
                        }
                        XR3_WSYNC();
                        np = 0;
                    };
                    for (int nt = 0; (vw >> 5) != 0u; nt++) {
                        const uint32_t pd = (vw >> 2) & 7u;
                        if (nt > N || pd > 5u) { status |= 0x100; break; }          // (distances strictly decrease: cannot happen)
                        if (doomable && (vw & 2u)) { doomed = true; d_held += 1; break; }
                        const int off = pd == 0u ? YZ : pd == 1u ? -Z : pd == 2u ? -YZ : pd == 3u ? Z : pd == 4u ? 1 : -1;
                        const int u = v + off;
                        const uint32_t uw = field[u];
                        const uint32_t step5 = (vw & ~31u) - (uw & ~31u) - ((vw & 2u) ? pen5 : 0u) - ((V2 && (vw & 1u)) ? guide5 : 0u);   // the edge itself, x32
                        if (vw & 2u) { d_vio += 1; d_held += 1; }
                        if (pd >= 4u) d_via += 1; else d_wl += (int)(step5 >> 5);
                        if (lane == 0) s_tmp[np] = (unsigned short)v;
                        plen++; np++;
                        fnv_mix(h, (uint32_t)gflat(v));
                        v = u; vw = uw;
                        if (np == XR3_TMP) { flush(); dirty = true; }
                    }
                    if (!doomed) flush();
                    if (doomed && lane == 0) s_retry = dirty ? 1 : 2;       // 2: the next attempt resumes at THIS search, 1: it starts over
                    if ((status & 0x100) || doomed) remaining = 0;
                    else {
                        // terminal node of the component: claimed (and recorded) only if nobody holds it yet
                        // (a source: not held and no owner bit <=> owner[v] == 0, see make_source; XR-Maze v2: the owner it will have)
                        if ((vw & 6u) == 0u) {
                            if (lane == 0) { if (!DEFER) owner[v] = (int16_t)a; if (plen < b.path_cap) path[plen] = gflat(v); field[v] = vw | 4u; }
                            plen++;
                            fnv_mix(h, (uint32_t)gflat(v));
                        }
                        remaining -= 1;
                        // the reached pin joins the component with all of its access points
                        for (int i = lane; i < nap; i += 64)
                            if (s_ap_pin[i] == (short)tpin) { s_ap_conn[i] = 1; make_source((uint32_t)s_ap_f[i], s_ap_own[i] != 0); }
                    }
                }
            }
            if (WIN && win_fail && lane == 0) s_fallback = 1;
            if (lane == 0) s_remaining = remaining;
        }
        xr_lds_barrier();      // every thread has left the round loop (its exit test reads s_min / s_bst) before the next search resets them
        XR_LAP(3);
    }
    // ---- does the attempt stand (XR-Maze v2)?  Its path uses a node held by another net and attempts are left: rip it up ----
    if (wv == sw) {
        const bool retry = V2 && b.maze_end_iter > 1 && d_held > 0 && attempt + 1 < b.maze_end_iter;
        if (DEFER && lane == 0 && !retry) s_retry = 0;          // (a doomed trace wrote 1 or 2: a held node is only ever met by one)
        if (!retry && lane == 0 && !(WIN && win_fail)) {
            if (n_isolated > 0) { d_vio += n_isolated; status |= XR_ENV_UNREACHABLE; }
#ifdef XR_PHASE_TIMING
            xr_step_epilogue(b, e, a, d_vio, d_wl, d_via, plen, status, nrounds, h, attempt + 1);       // (probe builds: XR_FETCH_TOUCHED = attempts)
#else
            xr_step_epilogue(b, e, a, d_vio, d_wl, d_via, plen, status, nrounds, h);
#endif
        }
    }
    XR_LAP(5);
    if (!DEFER) break;
    xr_lds_barrier();                                         // s_retry and the last sources are visible
    if (WIN && s_fallback) return false;                      // uniform: a certificate failed — nothing that counts was written
    const int retry_kind = s_retry;
    const bool retry = retry_kind != 0;
    if (!retry) {
        // ---- the attempt stands: its deferred claims.  A source word with the owner bit is a path node, the terminal node, or an
        // access point that had an owner when the route began.  Not held: owner 0 or this net -> this net (no load).  Held: another
        // net's wire (stays) or another net's unowned access point on the path (claimed, as XR-Maze v1 does): the owner is read.
        const int nq = (N + 3) >> 2;
        auto accept = [&](uint32_t w, int f) __attribute__((always_inline)) {
            if ((w >> 5) == 0u && (w & 4u)) {
                const int g = gflat(f);
                if (!(w & 2u)) owner[g] = (int16_t)a; else if (owner[g] == 0) owner[g] = (int16_t)a;
            }
        };
        for (int c = tid; c < nq; c += nthr) {
            const uint4 w = *reinterpret_cast<const uint4*>(field + (c << 2));
            const int f = c << 2;
            accept(w.x, f); accept(w.y, f + 1); accept(w.z, f + 2); accept(w.w, f + 3);
        }
        break;
    }
    // ---- rip-up: nothing reached global memory but the path list (overwritten by the next attempt).  The searches of this attempt
    // BEFORE the one whose path met a held node would repeat themselves exactly under the doubled penalty: their paths use no held
    // node, so they cost what they cost before while every alternative through a held node only got dearer — same distances along
    // them, same first tight predecessors, same targets.  The next attempt therefore RESUMES at the failed search:
    // the component (source words: access points of the connected pins, path nodes, their owner bits) and the tracing wave's sums
    // as they were when that search began; every other word goes back to "unreached" with its static bits.  LDS only.
    // (kind 1 — a trace so long that part of it was flushed before the held node showed: start over from the first pin.)
    attempt++;
    pen5 = ((uint32_t)b.pen_cost << 5) << attempt;
    resume = retry_kind == 2;
    for (int i = tid; i < mw; i += nthr) { s_open[i] = 0; s_defer[i] = 0; s_wmin[i] = XR_DIAL_INF; }
    if (!resume)
        for (int i = tid; i < nap; i += nthr) s_ap_conn[i] = (unsigned char)(s_ap_conn[i] == 2 ? 2 : (s_ap_pin[i] == (short)first_pin ? 1 : 0));
    xr_lds_barrier();
    {
        const int nq = (N + 3) >> 2;
        auto undo = [&](uint32_t w, int f) __attribute__((always_inline)) -> uint32_t {
            if (w == 0u) return 0u;
            if (resume && (w >> 5) == 0u) {                   // a source of the kept component: stays, and is open again
                uint32_t q, r;
                mask_pos((uint32_t)f, q, r);
                atomicOr(&s_open[r], 1u << q);
                s_wmin[r] = 0u;
                return w;
            }
            return XR3_UNREACHED_V2 | (w & 3u);
        };
        for (int c = tid; c < nq; c += nthr) {
            uint4* p4 = reinterpret_cast<uint4*>(field + (c << 2));
            uint4 w = *p4;
            const int f = c << 2;
            w.x = undo(w.x, f); w.y = undo(w.y, f + 1); w.z = undo(w.z, f + 2); w.w = undo(w.w, f + 3);
            *p4 = w;
        }
    }
    xr_lds_barrier();
    XR_LAP(7);
    }
    XR_TDUMP();
    return true;
}
