// xr_dial.h — XR-Maze v1 as a bucketed frontier expansion: the default router of the step kernels.
// Included by xr_kernels.hip only (device code, gfx950 / wave64).
//
// Why this form.  The distance field of a multi-source shortest-path problem is unique, and target choice and
// back-trace are deterministic functions of it, so any label-correcting order gives the oracle's result.  The
// line-segment sweeps of round 1 (xr_route_env, kept as the `XR_ROUTER_SWEEP` form for A/B) re-visit whole lines
// every iteration: 26 k wave-instructions per wave and 66 % of the wave-cycles parked at barriers / LDS waits
// (profiles/r02_a_route_sq_counters_r01_build.txt).  This router is work-efficient instead: Dial's algorithm on
// buckets of width `delta` (= dial_mult x the smallest edge length of the region), frontier kept as an `open` bitmask
// over the flat node index f = (x*Y + y)*Z + z (the reference observation's own order, so the field index IS the
// owner/node_net index and no layout transform is needed):
//
//   field[f]   u32   (distance << 2) | (held << 1) | 1      blockage: 0      unreached: 0xFFFFFFFD | held << 1
//   open       bit f: node lowered but not expanded yet
//   defer      bit f: an edge out of f was refused only because of the search bound; re-opened when the next search
//              of the same net resets the bound
//
// One round (one workgroup barrier): m = smallest open distance (accumulated during the previous round),
// every open node with d < m + delta is expanded — its (at most) 4 neighbours get `atomicMin(field, candidate)`;
// a neighbour that was lowered becomes open.  Open nodes beyond the bucket stay open.  The round also refreshes
// `best` = smallest tentative distance of an unconnected target; the search stops when m > best: every node with
// true distance < m is exact by then (label-correcting invariant: for every edge u->v, d(v) <= d(u) + w or u is
// open or deferred), which covers the target choice (ties included) and every node of the back-trace.  Candidates
// above `best` are not written (bound pruning, as in the sweep form).  A thread takes the open word it scans with an
// atomic exchange BEFORE it reads the distances, so a node lowered concurrently is either seen with its new value or
// re-opened by the thread that lowered it — no lost update for any bucket width.
//
// Buckets are keyed on f = d + h(v), not on d (A* order): h(v) = distance from v to the bounding box of the access points of the
// still unconnected pins (coordinate differences + via cost per layer: a consistent lower bound of the remaining cost, so f never
// decreases along a path and Dial's bucket order applies to f).  The search stops when the smallest open key exceeds `best`
// (the smallest tentative distance of a target, whose h is 0): every node with d + h <= best is exact by then — that covers every
// node on a shortest path to the chosen target, every tight predecessor the back-trace may look at, and every target tied
// with it.  Candidates with d + h > best are not written.  The explored set shrinks from a ball around the component to a band
// around the connection (6.5x fewer expansions on BASELINE config 3 regions, 2x on config 5; tools/sim/dial_sim.c), and all
// nodes of an obstacle-free straight connection share ONE key, so run-ahead carries a distance along it within one round.
//
// Later pins re-use the field: after a connection the path nodes and the reached pin's access points become sources
// (distance 0, open); every other value is still an upper bound.
//
// Two placements: LDS_FIELD (field + bitmasks in LDS: 38.2 KB at 24x40x9, 4 workgroups per CU) and the HBM-scratch form
// for large regions (see xr_dial_route_env_big below).
#pragma once
// XR-Maze v2's rip-up-and-reroute loop is computed as ONE attempt at the last attempt's penalty (proof: xr_dial3.h);
// -DXR3_V2_ALL_ATTEMPTS keeps round 4's attempt-by-attempt form (same results)
#ifdef XR3_V2_ALL_ATTEMPTS
#define XR3_ALL_ATTEMPTS true
#else
#define XR3_ALL_ATTEMPTS false
#endif


#define XR_DIAL_INF 0xFFFFFFFFu
#ifndef XR_DIAL_ASTAR
#define XR_DIAL_ASTAR 1          // LDS form: bucket keys f = d + h (0: plain Dijkstra order, keys = d) — A/B switch
#endif
#if defined(XR_PHASE_TIMING) && defined(XR_PROBE_SETUP)
#define XR_MARK(n) do { if ((n) == XR_PROBE_SETUP && threadIdx.x == XR_TIMING_TID) _ph[7] += clock64() - _t; } while (0)
#else
#define XR_MARK(n) do {} while (0)
#endif
#ifndef XR_BIG_SPEC_FLAGS
#define XR_BIG_SPEC_FLAGS 1     // HBM-scratch form: load node_net / owner of a neighbour together with its field word
#endif
#ifndef XR_POCKET_MASK
#define XR_POCKET_MASK 1        // LDS form: pocket flood with a bitmask as its visited set
#endif
#ifndef XR_DIAL_QUAD
#define XR_DIAL_QUAD 1         // LDS form: the nodes of a bucket are expanded by quads of lanes (one lane per direction)
#endif
#ifndef XR_SCAN_UNROLL
#define XR_SCAN_UNROLL 2      // open nodes of a mask word classified per loop iteration
#endif
#define XR_HB_MAX 6          // pin boxes of the search heuristic (HBM-scratch form; xr_dial3.h)
#define XR_QUAD_POOL 128       // nodes of a bucket queued per workgroup for the quads (split evenly over the waves); no room: next round
#ifndef XR_DIAL_CHAIN
#define XR_DIAL_CHAIN 1
#endif

// n / d and n % d with magic = floor(2^32 / d) (d >= 2) or 0xFFFFFFFF (d == 1); exact for n < 2^30
__device__ __forceinline__ void xr_divmod(uint32_t n, uint32_t d, uint32_t magic, uint32_t& q, uint32_t& r) {
    q = __umulhi(n, magic);
    r = n - q * d;
    if (r >= d) { q += 1; r -= d; }
}

// Node bitmasks use a TRANSPOSED bit order: node f lives in word f % mw, bit f / mw (mw = ceil(N / 32)).  A frontier is
// spatially clustered (consecutive f); this order deals its nodes round-robin over the words, i.e. over the lanes that
// scan them, so a round costs every lane about the same few bits instead of a few lanes a whole cluster.
__device__ __forceinline__ void xr_mask_or(uint32_t* mask, uint32_t f, uint32_t mw, uint32_t magic_mw) {
    uint32_t q, r;
    xr_divmod(f, mw, magic_mw, q, r);
    atomicOr(&mask[r], 1u << q);
}

__device__ __forceinline__ uint32_t xr_wave_min_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = __shfl_xor(v, off);
        v = o < v ? o : v;
    }
    return v;
}

// Prologue shared by both routers: auto-reset of a finished env, action validity (uniform over the workgroup).
// Returns true when net `a` is to be routed.
// (measured launch order, round 5) when the route began and which (region, net) it is: written by thread 0 of the prologue, read by the one
// thread that runs the epilogue — every router form has workgroup barriers in between
__shared__ long long xr_s_route_t0;
__shared__ int xr_s_route_net;
__device__ __forceinline__ bool xr_step_prologue(const XrBatchDev& b, const int e, const int a) {
    const int tid = threadIdx.x;
    // the three loads that depend on (e, a) only are issued together: a route is ~70 us and every dependent global round trip
    // in front of it costs 1-2 % of that
    const int nl0 = b.nlegal[e];
    const int r0 = b.env_region[e];
    const bool a_in_words = a >= 1 && ((a - 1) >> 6) < b.legal_words;
    const uint64_t lw0 = a_in_words ? b.legal[(int64_t)e * b.legal_words + ((a - 1) >> 6)] : 0ULL;
    if (nl0 == 0) {
        if (b.auto_reset) {
            xr_env_reset(b, e, 1, XR_ENV_WAS_RESET);
        } else if (tid == 0) {
            b.status[e] = XR_ENV_BAD_ACTION;
            b.delta[3 * e] = 0; b.delta[3 * e + 1] = 0; b.delta[3 * e + 2] = 0;
            b.reward[e] = -0.0; b.path_len[e] = 0; b.sweeps[e] = 0;
            xr_publish_record(b, e);
        }
        return false;
    }
    const bool valid = a_in_words && a <= b.regions[r0].n_nets && ((lw0 >> ((a - 1) & 63)) & 1ULL);
    if (valid && tid == 0 && b.net_meas) { xr_s_route_t0 = (long long)__builtin_readcyclecounter(); xr_s_route_net = b.regions[r0].net_off + a; }
    if (!valid) {   // the reference never checks this client-side; here: flagged no-op
        if (tid == 0) {
            b.status[e] = XR_ENV_BAD_ACTION;
            b.delta[3 * e] = 0; b.delta[3 * e + 1] = 0; b.delta[3 * e + 2] = 0;
            b.reward[e] = -0.0; b.path_len[e] = 0; b.sweeps[e] = 0;
            xr_publish_record(b, e);
        }
        return false;
    }
    return true;
}

// Game.step bookkeeping (reference baseline/baseline_utils.py:412, :426-438) + reward, by one thread.  The state it needs is
// loaded up front (one round trip), everything after that is a store — the packed record is written from registers, not read
// back from the arrays just written (that second round trip was 2 % of a route).
__device__ __forceinline__ void xr_step_epilogue(const XrBatchDev& b, const int e, const int a, int d_vio, int d_wl, int d_via,
                                                 int plen, int status, int nrounds, uint64_t h, int ntouched = 0) {
    if (plen > b.path_cap) status |= XR_ENV_PATH_TRUNC;
    uint64_t* lw = b.legal + (int64_t)e * b.legal_words + ((a - 1) >> 6);
    const int c0 = b.cum[3 * e + 0], c1 = b.cum[3 * e + 1], c2 = b.cum[3 * e + 2];
    const int nl = b.nlegal[e] - 1;
    const uint64_t lword = *lw;
    const int64_t steps = b.env_steps[e] + 1;
    // reward = -1 * (violation*500 + via*4 + wirelength*0.5)   (train_DQN.py:98-99), in double
    const double s = b.w_violation * (double)d_vio + b.w_via * (double)d_via + b.w_wirelength * (double)d_wl;
    b.cum[3 * e + 0] = c0 + d_vio; b.cum[3 * e + 1] = c1 + d_wl; b.cum[3 * e + 2] = c2 + d_via;
    b.delta[3 * e + 0] = d_vio; b.delta[3 * e + 1] = d_wl; b.delta[3 * e + 2] = d_via;
    b.reward[e] = -1.0 * s;
    *lw = lword & ~(1ULL << ((a - 1) & 63));
    b.nlegal[e] = nl;
    b.done[e] = (nl == 0);
    b.status[e] = status;
    b.path_len[e] = plen;
    b.sweeps[e] = nrounds;
    b.touched[e] = ntouched;
    if (b.net_meas) {          // what this route cost, for the launch orders of the next time this (region, net) is asked for
        const long long dt = (long long)__builtin_readcyclecounter() - xr_s_route_t0;
        const int cls = (int)min((dt >> b.meas_shift) + 1ll, 255ll);
        b.net_meas[xr_s_route_net] = (uint8_t)max(cls, 1);
#ifndef XR_PHASE_TIMING
        b.phase_cycles[(int64_t)e * 8 + 7] = dt;      // cycles of this env's last route (XR_FETCH_PHASES, slot 7): bench.py's launch utilisation = mean / max over a launch
#endif
    }
    fnv_mix(h, (uint32_t)a);
    fnv_mix(h, (uint32_t)d_vio); fnv_mix(h, (uint32_t)d_wl); fnv_mix(h, (uint32_t)d_via);
    fnv_mix(h, (uint32_t)plen);
    b.hash[e] = h;
    b.env_steps[e] = steps;
    atomicAdd(b.total_steps, 1ULL);
    XrStepRecord* r = b.records + e;              // (= xr_publish_record, from registers)
    r->reward = -1.0 * s;
    r->delta[0] = d_vio; r->delta[1] = d_wl; r->delta[2] = d_via;
    r->cum[0] = c0 + d_vio; r->cum[1] = c1 + d_wl; r->cum[2] = c2 + d_via;
    r->nlegal = nl;
    r->env_steps = (int32_t)steps;
    r->path_len = plen;
    r->done = (nl == 0); r->pad = 0;
    r->status = (uint16_t)status;
}


// ------------------------------------------------------------------------------------------------
// Isolated pins.  A pin whose access points sit in a pocket closed by BLOCKAGE nodes can never be reached (and, when it is
// the net's first pin, can never reach anything): XR-Maze v1 then charges one violation per unreachable pin — but only after a
// search with no reachable target has explored the WHOLE component (no bound ever prunes it), the slowest thing a router can be
// asked to do: 4-5 % of the BASELINE config 5 env-steps, each costing 10-40x an average route.  Pockets are tiny (1-3
// nodes in every case measured), so each wave floods one pin's pocket with a budget of XR_POCKET_CAP nodes before the first
// search: a flood that closes without meeting another pin's access point marks the pin isolated (s_ap_conn = 2) and the
// searches skip it.  Same results by construction (the pocket's boundary is static), nothing explored for nothing.
// ------------------------------------------------------------------------------------------------
#define XR_POCKET_CAP 12        // storage per wave (also the quads' node queue of the LDS form: do not shrink)
#ifndef XR_POCKET_BUDGET_LDS
#define XR_POCKET_BUDGET_LDS 12     // nodes a flood may visit before the pocket counts as open (<= XR_POCKET_CAP), LDS form
#endif
#define XR_POCKET_BUDGET_BIG 12     // ... HBM-scratch form (BASELINE config 5 has pockets of more than 4 nodes)
template <class ApT, class BlockedFn>
__device__ __forceinline__ void xr_mark_isolated_pins(int nap, const ApT* ap_f_of, const short* s_ap_pin, unsigned char* s_ap_conn,
                                                       int X, int Y, int Z, uint32_t ldir, uint32_t magic_yz, uint32_t magic_z,
                                                       BlockedFn blocked, int (*s_pocket)[XR_POCKET_CAP + 8], int* s_niso, int* s_src_iso,
                                                       int first_pin, const int budget) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nwv = min((int)(blockDim.x >> 6), 4);
    if (wv >= nwv) return;
    const int YZ = Y * Z;
    int* vis = s_pocket[wv];
    int pj = 0;                                           // running index of distinct pins (wave-uniform)
    for (int i0 = 0; i0 < nap; i0++) {
        if (!(s_ap_conn[i0] & 0x80)) continue;           // (bit 7: first access point of its pin, set by the caller)
        const short pin = s_ap_pin[i0];
        const int mine = (pj++ % nwv) == wv;
        if (!mine) continue;
        // seed: the pin's access points
        int cnt = 0;
        bool open_pocket = false;
        for (int b0 = 0; b0 < nap; b0 += 64) {
            const int i = b0 + lane;
            const bool is = i < nap && s_ap_pin[i] == pin;
            const unsigned long long mm = __ballot(is);
            const int pos = cnt + __popcll(mm & ((1ULL << lane) - 1ULL));
            if (is && pos < budget) vis[pos] = (int)ap_f_of[i];
            cnt += __popcll(mm);
        }
        if (cnt > budget) continue;                // (not a small pocket)
        __builtin_amdgcn_wave_barrier();
        int lo = 0;
        while (lo < cnt && !open_pocket) {
            const int hi = min(cnt, lo + 16);
            // lane = (frontier node, direction)
            const int k = lo + (lane >> 2), dir = lane & 3;
            int nf = -1;
            if (k < hi) {
                const int f = vis[k];
                uint32_t x, r, y, z;
                xr_divmod((uint32_t)f, (uint32_t)YZ, magic_yz, x, r);
                xr_divmod(r, (uint32_t)Z, magic_z, y, z);
                const bool vert = (ldir >> z) & 1u;
                if (dir == 0) nf = vert ? ((int)y + 1 < Y ? f + Z : -1) : ((int)x + 1 < X ? f + YZ : -1);
                else if (dir == 1) nf = vert ? (y > 0 ? f - Z : -1) : (x > 0 ? f - YZ : -1);
                else if (dir == 2) nf = ((int)z + 1 < Z) ? f + 1 : -1;
                else nf = (z > 0) ? f - 1 : -1;
                if (nf >= 0 && blocked(nf)) nf = -1;
                if (nf >= 0) for (int q = 0; q < cnt; q++) if (vis[q] == nf) { nf = -1; break; }      // (cnt <= budget)
            }
            unsigned long long cand = __ballot(nf >= 0);
            lo = hi;
            while (cand) {                                // append in lane order, de-duplicating against what was just appended
                const int src = __ffsll((long long)cand) - 1;
                cand &= cand - 1;
                const int c = __builtin_amdgcn_readlane(nf, src);
                if (__ballot(lane < cnt && vis[lane] == c)) continue;          // already in the pocket (cnt <= 64 entries)
                if (cnt >= budget) { open_pocket = true; break; }
                if (lane == 0) vis[cnt] = c;
                cnt++;
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (open_pocket) continue;
        // closed pocket: isolated unless it holds an access point of another pin of the net
        bool other = false;
        for (int b0 = 0; b0 < nap; b0 += 64) {
            const int i = b0 + lane;
            bool hit = false;
            if (i < nap && s_ap_pin[i] != pin) { const int f = (int)ap_f_of[i]; for (int q = 0; q < cnt; q++) hit |= (vis[q] == f); }
            if (__ballot(hit)) other = true;
        }
        if (other) continue;
        for (int i = lane; i < nap; i += 64) if (s_ap_pin[i] == pin) s_ap_conn[i] = (s_ap_conn[i] & 0x80) | 2;
        if (lane == 0) { if ((int)pin == first_pin) *s_src_iso = 1; else atomicAdd(s_niso, 1); }
    }
}

// ------------------------------------------------------------------------------------------------
// XR-Maze v2 guide of the net being routed, in LDS: up to XR_GUIDE_MAX_BOXES boxes packed as x0 | x1 << 16, y0 | y1 << 16,
// z0 | z1 << 16 (margin applied, clamped to 0..0x7FFF).  The region's own boxes for the net when it carries any
// (xr_batch_load_guides: the global-route guide rectangles clipped to the region), else ONE box: the bounding box
// s_bb = {x0, x1, y0, y1} of the net's access points on every layer.  Every thread calls xr_guide_load after s_bb is complete
// and visible; a barrier must follow.  
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void xr_guide_load(const XrBatchDev& b, const XrRegionDev& R, int a, const int* s_bb, int Z,
                                              int4* s_gbx, int* s_ngb, int tid) {
    int lo = 0, hi = 0;
    if (b.guide_csr) { lo = b.guide_csr[R.net_off + a]; hi = b.guide_csr[R.net_off + a + 1]; }
    const int n = hi - lo, m = b.guide_margin;
    auto pack = [](int c0, int c1) { return (int)((uint32_t)max(c0, 0) | ((uint32_t)min(c1, 0x7FFF) << 16)); };
    if (n > 0) {
        if (tid < n) {
            const int16_t* __restrict__ g = b.guide_box + 6 * (int64_t)(lo + tid);
            s_gbx[tid] = make_int4(pack(g[0] - m, g[2] + m), pack(g[1] - m, g[3] + m), pack(g[4], g[5]), 0);
        }
        if (tid == 0) *s_ngb = n;
    } else if (tid == 0) {
        s_gbx[0] = make_int4(pack(s_bb[0] - m, s_bb[1] + m), pack(s_bb[2] - m, s_bb[3] + m), pack(0, Z - 1), 0);
        *s_ngb = 1;
    }
}
__device__ __forceinline__ bool xr_guide_in(const int4 g, int x, int y, int z) {
    return x >= (g.x & 0xFFFF) && x <= (int)((uint32_t)g.x >> 16) && y >= (g.y & 0xFFFF) && y <= (int)((uint32_t)g.y >> 16) &&
           z >= (g.z & 0xFFFF) && z <= (int)((uint32_t)g.z >> 16);
}
// g0 = box 0 in (scalar) registers: the common case of one box costs no LDS read
__device__ __forceinline__ bool xr_guide_has(const int4* s_gbx, int n, const int4 g0, int x, int y, int z) {
    bool in = xr_guide_in(g0, x, y, z);
    for (int i = 1; i < n; i++) in = in || xr_guide_in(s_gbx[i], x, y, z);
    return in;
}

// ------------------------------------------------------------------------------------------------
// LDS form.  LDS carve (dynamic):  field u32[n_max] | open | defer | claim | wmin  (u32[mw_max] each, mw_max =
// n_max / 32 + 1) | el4x | el4y.
//   wmin[w]  lower bound of the distances of the open nodes of word w (XR_DIAL_INF: none): a round reads ONE word per
//            lane to know whether any of its ~32 nodes falls into the bucket; only those words are taken and classified.
// ------------------------------------------------------------------------------------------------
// V2: XR-Maze v2 knobs compiled in (guide cost, rip-up-and-reroute); the default instantiation carries none of their cost
template <bool V2>
__device__ __forceinline__ void xr_dial_route_env(const XrBatchDev& b, const int e, const int a, char* smem) {
    __shared__ unsigned short s_ap_f[XR_MAX_AP_PER_NET];      // flat node index (the LDS form holds < 64 k nodes)
    __shared__ short s_ap_pin[XR_MAX_AP_PER_NET];
    __shared__ unsigned char s_ap_conn[XR_MAX_AP_PER_NET];
    __shared__ uint32_t s_min[3], s_bst[3];
    __shared__ unsigned long long s_tkey;
    __shared__ int s_hb[6];                                   // bounding box of the unconnected targets: x, y (coordinates x4), z
    __shared__ int s_remaining, s_target_i, s_first_pin, s_npins, s_niso, s_src_iso;
    __shared__ int s_pocket[4][XR_POCKET_CAP + 8];
    __shared__ int s_gb[4], s_retry, s_ngb;                   // XR-Maze v2: bounding box of the net's access points (track indices), rip-up decision
    __shared__ int4 s_gbx[XR_GUIDE_MAX_BOXES];                // ... and its guide (xr_guide_load)
    __shared__ int s_abort;                                   // a hop loop hit the round cap

    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    if (!xr_step_prologue(b, e, a)) return;

    XR_T0();
    const XrRegionDev R = b.regions[b.env_region[e]];
    const int X = R.X, Y = R.Y, Z = R.Z, N = R.N;
    const int YZ = Y * Z;
    const uint32_t ldir = R.ldir_mask;
    const int mw = (N + 31) >> 5;
    const int round_cap = b.round_cap > 0 ? b.round_cap : 1024 + N;
    const uint32_t umw = (uint32_t)mw, magic_mw = R.magic_mw;
    const int16_t* __restrict__ node_net = b.rg_node_net + R.node_off;
    int16_t* __restrict__ owner = b.owner + (int64_t)e * b.n_max;

    uint32_t* field = reinterpret_cast<uint32_t*>(smem);
    const int mw_max = (b.n_max >> 5) + 1;
    uint32_t* s_open = field + b.n_max;
    uint32_t* s_defer = s_open + mw_max;
    uint32_t* s_claim = s_defer + mw_max;
    uint32_t* s_wmin = s_claim + mw_max;
    // coordinate tables (x4, relative to the first track): s_xc[k] = 4*(xs[clamp(k-1)] - xs[0]), k = 0 .. X+1, so that the
    // coordinate of track x is s_xc[x+1] and the edge between tracks i-1 and i is s_xc[i+1] - s_xc[i] (0 at both ends)
    uint32_t* s_xc = s_wmin + mw_max;
    uint32_t* s_yc = s_xc + (b.x_max + 2);

    // access points of the net and the edge tables: loads issued now, consumed after the grid build (three dependent
    // global round trips otherwise sit on the critical path of a 100 us kernel)
    const int ap_lo = b.net_csr[R.net_off + a], ap_hi = b.net_csr[R.net_off + a + 1];
    const int nap = ap_hi - ap_lo;    // 1 <= nap <= XR_MAX_AP_PER_NET (checked at load)
    int my_ap_f = 0, my_ap_pin = 0;
    if (tid < nap) { my_ap_f = b.ap_node[R.ap_off + ap_lo + tid]; my_ap_pin = b.ap_pin[R.ap_off + ap_lo + tid]; }
    uint32_t my_xc = 0, my_yc = 0;
    if (tid <= X + 1) my_xc = (uint32_t)(b.coords[R.xs_off + min(max(tid - 1, 0), X - 1)] - b.coords[R.xs_off]) << 2;
    if (tid <= Y + 1) my_yc = (uint32_t)(b.coords[R.ys_off + min(max(tid - 1, 0), Y - 1)] - b.coords[R.ys_off]) << 2;

    // ---- grid build: field word of every node for THIS net.  node_net / owner rows are padded to multiples of 8
    // elements (16-byte loads); pad slots and blockages become 0.  Loads of four chunks are issued before the first use.
    // (a lambda: XR-Maze v2 rebuilds the field for every rip-up-and-reroute attempt)
    auto build_field = [&]() {
    {
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
                    const uint32_t ww = XR_W_UNREACHED | (((ow != 0 && ow != a) || (nn > 0 && nn != a)) ? 2u : 0u);
                    w[j] = (nn == -1 || f0 + j >= N) ? XR_W_BLOCK : ww;
                }
                uint4* dst = reinterpret_cast<uint4*>(field + f0);
                dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
                dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
            }
        }
    }
        for (int i = tid; i < mw; i += nthr) { s_open[i] = 0; s_defer[i] = 0; s_claim[i] = 0; s_wmin[i] = XR_DIAL_INF; }
    };
    XR_MARK(1);
    build_field();
    XR_MARK(2);
    if (tid <= X + 1) s_xc[tid] = my_xc;
    if (tid <= Y + 1) s_yc[tid] = my_yc;
    for (int i = tid + nthr; i <= X + 1; i += nthr)
        s_xc[i] = (uint32_t)(b.coords[R.xs_off + min(i - 1, X - 1)] - b.coords[R.xs_off]) << 2;
    for (int i = tid + nthr; i <= Y + 1; i += nthr)
        s_yc[i] = (uint32_t)(b.coords[R.ys_off + min(i - 1, Y - 1)] - b.coords[R.ys_off]) << 2;
    auto el4x = [&](int i) { return s_xc[i + 1] - s_xc[i]; };
    auto el4y = [&](int i) { return s_yc[i + 1] - s_yc[i]; };
    (void)el4x; (void)el4y;
    if (tid == 0) {
        s_first_pin = 0x7FFFFFFF; s_npins = 0; s_niso = 0; s_src_iso = 0; s_abort = 0;
        s_gb[0] = 0x7FFFFFFF; s_gb[1] = -1; s_gb[2] = 0x7FFFFFFF; s_gb[3] = -1;
    }
    __syncthreads();
    XR_MARK(3);
    for (int i = tid; i < nap; i += nthr) {
        const int pin = i < nthr ? my_ap_pin : (int)b.ap_pin[R.ap_off + ap_lo + i];
        const int apf = i < nthr ? my_ap_f : b.ap_node[R.ap_off + ap_lo + i];
        s_ap_f[i] = (unsigned short)apf;
        s_ap_pin[i] = (short)pin;
        s_ap_conn[i] = 0;
        atomicMin(&s_first_pin, pin);
        if (V2 && b.guide_cost) {                // XR-Maze v2: the net's guide = bounding box of all its access points (+ margin)
            const int gy = (apf / Z) % Y, gx = apf / YZ;
            atomicMin(&s_gb[0], gx); atomicMax(&s_gb[1], gx); atomicMin(&s_gb[2], gy); atomicMax(&s_gb[3], gy);
        }
    }
    __syncthreads();
    // a node becomes a source: distance 0, open
    auto make_source = [&](uint32_t f) {
        uint32_t q, r;
        xr_divmod(f, umw, magic_mw, q, r);
        field[f] &= 3u;
        atomicOr(&s_open[r], 1u << q);
        s_wmin[r] = 0u;                       // (racing plain stores of the same value)
    };
    // component = all access points of the lowest pin id; number of distinct pins
    for (int i = tid; i < nap; i += nthr) {
        const short pin = s_ap_pin[i];
        bool seen = false;
        for (int j = 0; j < i; j++) seen |= (s_ap_pin[j] == pin);
        if (!seen) atomicAdd(&s_npins, 1);
        unsigned char cflag = seen ? 0 : 0x80;            // bit 7: first access point of its pin (for xr_mark_isolated_pins)
        if (pin == (short)s_first_pin) { cflag |= 1; make_source(s_ap_f[i]); }
        s_ap_conn[i] = cflag;
    }
    __syncthreads();
    XR_MARK(4);
    // pins in closed pockets are never searched for (see xr_mark_isolated_pins)
#if XR_POCKET_MASK
    // LDS form: the flood keeps its visited set in a node bitmask that is still all-zero at this point (wave 0: defer, wave 1:
    // claim) — test-and-set by one LDS atomic per candidate, all candidates of a pass at once — instead of comparing every
    // candidate with every node visited so far; the bits are cleared again afterwards.  Same pockets, same verdicts.
    {
        const int lane = tid & 63, wv = tid >> 6;
        if (wv < 2) {
            uint32_t* vmask = wv == 0 ? s_defer : s_claim;
            int* pcnt = &s_pocket[wv][0];
            unsigned short* vis = reinterpret_cast<unsigned short*>(&s_pocket[wv][1]);
            const int vcap = 2 * (XR_POCKET_CAP + 8 - 1);
            const int nwv = min((int)(nthr >> 6), 2);
            auto visit = [&](uint32_t f) {                 // new in this flood: counted and (room permitting) listed
                uint32_t q, r;
                xr_divmod(f, umw, magic_mw, q, r);
                const uint32_t bit = 1u << q;
                if (!(atomicOr(&vmask[r], bit) & bit)) {
                    const int pos = atomicAdd(pcnt, 1);
                    if (pos < vcap) vis[pos] = (unsigned short)f;
                    else atomicAnd(&vmask[r], ~bit);      // (far beyond the budget: the pocket is open anyway)
                }
            };
            int pj = 0;
            for (int i0 = 0; i0 < nap; i0++) {
                if (!(s_ap_conn[i0] & 0x80)) continue;    // (bit 7: first access point of its pin)
                const short pin = s_ap_pin[i0];
                if ((pj++ % nwv) != wv) continue;
                if (lane == 0) *pcnt = 0;
                __builtin_amdgcn_wave_barrier();
                for (int i = lane; i < nap; i += 64) if (s_ap_pin[i] == pin) visit((uint32_t)s_ap_f[i]);
                __builtin_amdgcn_wave_barrier();
                int cnt = __builtin_amdgcn_readfirstlane(*pcnt), lo = 0;
                while (lo < cnt && cnt <= XR_POCKET_BUDGET_LDS) {
                    const int hi = min(cnt, lo + 16);
                    const int k = lo + (lane >> 2), dir = lane & 3;
                    if (k < hi) {
                        const int f = (int)vis[k];
                        uint32_t x, r, y, z;
                        xr_divmod((uint32_t)f, (uint32_t)YZ, R.magic_yz, x, r);
                        xr_divmod(r, (uint32_t)Z, R.magic_z, y, z);
                        const bool vert = (ldir >> z) & 1u;
                        const int sgn = (dir & 1) ? -1 : 1;
                        const bool planar = dir < 2;
                        const int ddx = (planar && !vert) ? sgn : 0, ddy = (planar && vert) ? sgn : 0, ddz = planar ? 0 : sgn;
                        const int nx = (int)x + ddx, ny = (int)y + ddy, nz = (int)z + ddz;
                        if ((unsigned)nx < (unsigned)X && (unsigned)ny < (unsigned)Y && (unsigned)nz < (unsigned)Z) {
                            const int nf = f + ddx * YZ + ddy * Z + ddz;
                            if (field[nf] != XR_W_BLOCK) visit((uint32_t)nf);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    lo = hi;
                    cnt = __builtin_amdgcn_readfirstlane(*pcnt);
                }
                const bool open_pocket = cnt > XR_POCKET_BUDGET_LDS;
                // closed pocket: isolated unless it holds an access point of another pin of the net
                bool other = false;
                if (!open_pocket)
                    for (int b0 = 0; b0 < nap; b0 += 64) {
                        const int i = b0 + lane;
                        bool hit = false;
                        if (i < nap && s_ap_pin[i] != pin) {
                            uint32_t q, r;
                            xr_divmod((uint32_t)s_ap_f[i], umw, magic_mw, q, r);
                            hit = (vmask[r] >> q) & 1u;
                        }
                        if (__ballot(hit)) other = true;
                    }
                __builtin_amdgcn_wave_barrier();
                for (int i = lane; i < min(cnt, vcap); i += 64) {          // the mask goes back to all-zero
                    uint32_t q, r;
                    xr_divmod((uint32_t)vis[i], umw, magic_mw, q, r);
                    atomicAnd(&vmask[r], ~(1u << q));
                }
                __builtin_amdgcn_wave_barrier();
                if (open_pocket || other) continue;
                for (int i = lane; i < nap; i += 64) if (s_ap_pin[i] == pin) s_ap_conn[i] = (s_ap_conn[i] & 0x80) | 2;
                if (lane == 0) { if ((int)pin == s_first_pin) s_src_iso = 1; else atomicAdd(&s_niso, 1); }
            }
        }
    }
#else
    xr_mark_isolated_pins(nap, s_ap_f, s_ap_pin, s_ap_conn, X, Y, Z, ldir, R.magic_yz, R.magic_z,
                          [&](int f) { return field[f] == XR_W_BLOCK; }, s_pocket, &s_niso, &s_src_iso, s_first_pin, XR_POCKET_BUDGET_LDS);
#endif
    __syncthreads();
    XR_MARK(5);
    for (int i = tid; i < nap; i += nthr) s_ap_conn[i] &= 0x7F;
    const int n_isolated = s_src_iso ? s_npins - 1 : s_niso;       // unreachable pins known up front
    if (tid == 0) s_remaining = s_npins - 1 - n_isolated;
    XR_LAP(0);

    const uint32_t via4 = (uint32_t)b.via_cost << 2;
    uint32_t pen4 = (uint32_t)b.pen_cost << 2;              // (XR-Maze v2: doubled by every rip-up-and-reroute attempt)
    const uint32_t delta = R.w_min * (uint32_t)b.dial_mult;
    const uint32_t uYZ = (uint32_t)YZ, uZ = (uint32_t)Z;
    int d_vio = 0, d_wl = 0, d_via = 0, plen = 0, status = XR_ENV_OK, d_held = 0;   // thread 0 only
    int nrounds = 0;
    const uint64_t h0 = (tid == 0) ? b.hash[e] : 0;
    uint64_t h = h0;
    int32_t* __restrict__ path = b.path + (int64_t)e * b.path_cap;
    // XR-Maze v2, neutral by default.  Guide: entering a node outside the net's guide box costs guide_cost.
    const uint32_t guide4 = V2 ? (uint32_t)b.guide_cost << 2 : 0u;
    if (V2 && b.guide_cost) {
        xr_guide_load(b, R, a, s_gb, Z, s_gbx, &s_ngb, tid);
        __syncthreads();
    }
    const int ngb = (V2 && b.guide_cost) ? s_ngb : 0;
    int4 gb0 = make_int4(0, 0, 0, 0);
    if (V2 && b.guide_cost) {
        const int4 t = s_gbx[0];
        gb0 = make_int4(__builtin_amdgcn_readfirstlane(t.x), __builtin_amdgcn_readfirstlane(t.y), __builtin_amdgcn_readfirstlane(t.z), 0);
    }
    auto guide_of = [&](int x, int y, int z) -> uint32_t {
        if (!V2) return 0u;
        return (guide4 != 0u && !xr_guide_has(s_gbx, ngb, gb0, x, y, z)) ? guide4 : 0u;
    };
    // Rip-up and reroute: claims of an attempt are tentative (owner = -a) until the attempt stands
    const int16_t claim_val = (int16_t)(XR3_ALL_ATTEMPTS && V2 && b.maze_end_iter > 1 ? -a : a);      // (one attempt: its claims are final)
#ifdef XR3_V2_ALL_ATTEMPTS
    int attempt = 0;
#else           // the rip-up-and-reroute loop has one possible outcome: the last attempt's route (xr_dial3.h)
    int attempt = V2 ? max(b.maze_end_iter, 1) - 1 : 0;
    if (V2) pen4 <<= attempt;
#endif

    for (;;) {                                              // attempts (exactly one unless maze_end_iter > 1)
    for (;;) {
        // ---- new search: sources are open with distance 0; deferred nodes are looked at again ------------
        if (tid == 0) {
            s_min[0] = 0; s_min[1] = XR_DIAL_INF; s_min[2] = XR_DIAL_INF;
            s_bst[0] = XR_DIAL_INF; s_bst[1] = XR_DIAL_INF; s_bst[2] = XR_DIAL_INF;
            s_tkey = ~0ULL;
            s_hb[0] = 0x7FFFFFFF; s_hb[1] = -0x7FFFFFFF; s_hb[2] = 0x7FFFFFFF; s_hb[3] = -0x7FFFFFFF; s_hb[4] = 0x7FFFFFFF; s_hb[5] = -1;
        }
        for (int i = tid; i < mw; i += nthr) {
            const uint32_t m = s_defer[i];
            if (m) { atomicOr(&s_open[i], m); s_defer[i] = 0; s_wmin[i] = 0u; }   // (0: a lower bound; the first scan fixes it)
        }
        __syncthreads();
        if (s_remaining <= 0) break;          // uniform: written before the barrier above
        // heuristic of this search: bounding box of the access points of the unconnected (and not isolated) pins
        for (int i = tid; i < nap; i += nthr)
            if (!s_ap_conn[i]) {
                uint32_t ax, ar, ay, az;
                xr_divmod((uint32_t)s_ap_f[i], uYZ, R.magic_yz, ax, ar);
                xr_divmod(ar, uZ, R.magic_z, ay, az);
                atomicMin(&s_hb[0], (int)s_xc[ax + 1]); atomicMax(&s_hb[1], (int)s_xc[ax + 1]);
                atomicMin(&s_hb[2], (int)s_yc[ay + 1]); atomicMax(&s_hb[3], (int)s_yc[ay + 1]);
                atomicMin(&s_hb[4], (int)az); atomicMax(&s_hb[5], (int)az);
            }
        __syncthreads();
        const int hb0 = s_hb[0], hb1 = s_hb[1], hb2 = s_hb[2], hb3 = s_hb[3], hb4 = s_hb[4], hb5 = s_hb[5];
        // h(v): distance to that box — coordinate differences + one via cost per layer (a consistent lower bound)
        auto heur_c = [&](int xc, int yc, int z) -> uint32_t {                 // (from coordinates x4)
            if (!XR_DIAL_ASTAR) return 0u;
            const int hx = max(0, max(hb0 - xc, xc - hb1)), hy = max(0, max(hb2 - yc, yc - hb3));
            const int hz = max(0, max(hb4 - z, z - hb5));
            return ((uint32_t)(hx + hy) >> 2) + (uint32_t)hz * (uint32_t)b.via_cost;
        };
        auto heur = [&](int x, int y, int z) -> uint32_t { return heur_c((int)s_xc[x + 1], (int)s_yc[y + 1], z); };
        (void)heur;
        int cur = 0;
        // Bounded: a search takes a few tens of rounds (BASELINE configs: <= 43); round_cap = 1024 + N is far beyond any legal
        // region (a maze whose only path visits every node needs ~N/4).  Past it the router gives up on this net
        // (XR_ENV_ROUTER_ABORT): the env, not the device, pays for a pathological input.
        bool aborted = false;
        for (int nsr = 0;; nsr++) {
            const int nx1 = cur == 2 ? 0 : cur + 1, nx2 = nx1 == 2 ? 0 : nx1 + 1;
            const uint32_t m = s_min[cur], best = s_bst[cur];
            if (m == XR_DIAL_INF || m > best) break;                 // uniform
            if (nsr >= round_cap || s_abort) { aborted = true; break; }   // uniform (s_abort: written before the last barrier)
            const uint32_t hi = m + delta;
            uint32_t lmin = XR_DIAL_INF;
            if (tid == 0) { s_min[nx2] = XR_DIAL_INF; s_bst[nx2] = XR_DIAL_INF; }
            // bound for the next round: smallest tentative distance of an unconnected target
            // (by the threads at the END of the workgroup: the first wave carries the words beyond one per thread)
            for (int i = nthr - 1 - tid; i < nap; i += nthr)
                if (!s_ap_conn[i]) { const uint32_t w = field[s_ap_f[i]]; if (w < XR_W_UNREACHED) atomicMin(&s_bst[nx1], w >> 2); }
            // A lane scans word wi and — where the mask has more words than the workgroup has threads — word wi + nthr in
            // the SAME pass (one 64-bit bit set), so that no wave runs the body twice per round.
#if XR_DIAL_QUAD
            // A lane scans word wi and — where the mask has more words than the workgroup has threads — word wi + nthr in
            // the SAME pass (one 64-bit bit set).  The nodes of this bucket are then expanded by QUADS of lanes, one lane per
            // direction: a hop of the run-ahead chain is one neighbour's worth of instructions instead of four (a route is
            // a sequential chain of hops, and a hop done by one lane costs ~2 k cycles of instruction issue even on an idle chip).
            for (int wbase = 0; wbase < mw; wbase += 2 * nthr) {          // (uniform trip count: the expansion is wave-cooperative)
                const int wi = wbase + tid;
                const int wi2 = wi + nthr;
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
                            // (loads from safe addresses instead of guarded ones, all issued before the first use: one round trip
                            //  for the distances, one for the coordinates)
                            int fq[XR_SCAN_UNROLL];
                            uint32_t cx[XR_SCAN_UNROLL], cy[XR_SCAN_UNROLL], cz[XR_SCAN_UNROLL], xcq[XR_SCAN_UNROLL], ycq[XR_SCAN_UNROLL];
#pragma unroll
                            for (int j = 0; j < XR_SCAN_UNROLL; j++) fq[j] = q[j] >= 0 ? (q[j] & 31) * mw + (q[j] < 32 ? wi : wi2) : 0;
#pragma unroll
                            for (int j = 0; j < XR_SCAN_UNROLL; j++) w[j] = field[fq[j]];
#pragma unroll
                            for (int j = 0; j < XR_SCAN_UNROLL; j++) {
                                uint32_t cr;
                                xr_divmod((uint32_t)fq[j], uYZ, R.magic_yz, cx[j], cr);
                                xr_divmod(cr, uZ, R.magic_z, cy[j], cz[j]);
                            }
#pragma unroll
                            for (int j = 0; j < XR_SCAN_UNROLL; j++) { xcq[j] = s_xc[cx[j] + 1]; ycq[j] = s_yc[cy[j] + 1]; }
#pragma unroll
                            for (int j = 0; j < XR_SCAN_UNROLL; j++) {
                                if (q[j] < 0) continue;
                                const uint32_t key = (w[j] >> 2) + heur_c((int)xcq[j], (int)ycq[j], (int)cz[j]);
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
                    const int lane = tid & 63, wv = tid >> 6;
                    const int qcap = XR_QUAD_POOL / ((nthr + 63) >> 6);
                    int* qcnt = &s_pocket[0][0] + wv;                                                   // [16] counters
                    unsigned short* qn = reinterpret_cast<unsigned short*>(&s_pocket[0][0] + 16) + wv * qcap;     // [XR_QUAD_POOL] nodes
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
                    // ---- quads: lanes 4g .. 4g+3 follow ONE chain, lane 4g+d relaxes direction d of the chain's current node
                    const int dir = lane & 3, qbase = lane & ~3;
                    int gf = -1, gx = 0, gy = 0, gz = 0, qh = 0;
                    uint32_t gd4 = 0;
                    for (int nhop = 0;; nhop++) {
                        if (nhop >= round_cap) {          // (every hop lowers a field word: finite anyway; the cap is the hang guard)
                            if (gf >= 0 && dir == 0) { xr_mask_or(s_open, (uint32_t)gf, umw, magic_mw); s_abort = 1; }
                            break;
                        }
                        const unsigned long long idle_g = __ballot(gf < 0) & 0x1111111111111111ULL;       // one bit per idle quad
                        if (gf < 0) {
                            const int idx = qh + __popcll(idle_g & ((1ULL << qbase) - 1ULL));
                            if (idx < nq) {
                                gf = (int)qn[idx];
                                uint32_t ux, ur, uy, uz;
                                xr_divmod((uint32_t)gf, uYZ, R.magic_yz, ux, ur);
                                xr_divmod(ur, uZ, R.magic_z, uy, uz);
                                gx = (int)ux; gy = (int)uy; gz = (int)uz;
                            }
                        }
                        qh += __popcll(idle_g);
                        if (__ballot(gf >= 0) == 0ULL) break;                                            // uniform
#ifdef XR_COUNT_HOPS
                        if (tid == XR_TIMING_TID) _ph[7] += 1;          // (probe: hop iterations of this wave instead of rounds)
#endif
                        // ---- one hop of every active quad.  All LDS reads are issued together (addresses made safe instead of
                        // guarded), then pure ALU, then the one atomic: two LDS round trips per hop
                        const bool act = gf >= 0;
                        const int gfs = act ? gf : 0;
                        const bool vert = (ldir >> gz) & 1u;
                        const int sgn = (dir & 1) ? -1 : 1;
                        const bool planar = dir < 2;
                        const int ddx = (planar && !vert) ? sgn : 0, ddy = (planar && vert) ? sgn : 0, ddz = planar ? 0 : sgn;
                        const int nx = gx + ddx, ny = gy + ddy, nz = gz + ddz;
                        const bool inb = act && (unsigned)nx < (unsigned)X && (unsigned)ny < (unsigned)Y && (unsigned)nz < (unsigned)Z;
                        const int nf = inb ? gfs + ddx * YZ + ddy * Z + ddz : gfs;
                        // coordinates x4 (tables: track i at [i + 1], padded at both ends): own x, own y, neighbour's along the layer's axis
                        const uint32_t gw = field[gfs], wn = field[nf];
                        const uint32_t xq = s_xc[gx + 1], yq = s_yc[gy + 1];
                        const uint32_t cb = (vert ? s_yc : s_xc)[(vert ? gy : gx) + 1 + (planar ? sgn : 0)];
                        const uint32_t ca = vert ? yq : xq;
                        const uint32_t len4 = planar ? (sgn > 0 ? cb - ca : ca - cb) : via4;
                        gd4 = gw & ~3u;
                        const uint32_t cand4 = gd4 + len4 + ((wn & 2u) ? pen4 : 0u) + guide_of(nx, ny, nz);
                        const uint32_t cw = cand4 | (wn & 3u);
                        const uint32_t key = (cand4 >> 2) + heur_c((int)((planar && !vert) ? cb : xq), (int)((planar && vert) ? cb : yq), nz);   // f = d + h
                        // blockage, a distance that does not exist (>= XR_DIST_CAP, spec), or no improvement: nothing to do
                        const bool go = inb && wn != XR_W_BLOCK && cand4 < XR_W_USABLE_END && cw < wn;
                        const bool refused = go && key > best;                             // bound pruning (on f)
                        bool lowered = false, chain_cand = false;
                        if (go && !refused) {
                            const uint32_t old = atomicMin(&field[nf], cw);
                            lowered = cw < old;
                            chain_cand = lowered && XR_DIAL_CHAIN && key < hi;
                        }
                        const uint32_t c4 = (uint32_t)(__ballot(chain_cand) >> qbase) & 15u;
                        const uint32_t r4 = (uint32_t)(__ballot(refused) >> qbase) & 15u;
                        const int win = c4 ? __ffs((int)c4) - 1 : -1;                   // the chain goes on with the first lowered direction
                        if (lowered && dir != win) {                                    // the others become open
                            uint32_t oq, orr;
                            xr_divmod((uint32_t)nf, umw, magic_mw, oq, orr);
                            atomicOr(&s_open[orr], 1u << oq);
                            atomicMin(&s_wmin[orr], key);
                            lmin = key < lmin ? key : lmin;
                        }
                        if (gf >= 0) {
                            if (r4 && dir == 0) xr_mask_or(s_defer, (uint32_t)gf, umw, magic_mw);
                            if (win >= 0) {                                 // every lane of the quad moves to the winner's node
                                const int sw = (win & 1) ? -1 : 1;
                                const int wx = (win < 2 && !vert) ? sw : 0, wy = (win < 2 && vert) ? sw : 0, wz = win < 2 ? 0 : sw;
                                gf += wx * YZ + wy * Z + wz;
                                gx += wx; gy += wy; gz += wz;
                            } else gf = -1;
                        }
                    }
                }
                XR_LAP(2);
            }
#else
            for (int wi = tid; wi < mw; wi += 2 * nthr) {
                const int wi2 = wi + nthr;
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
                    unsigned long long keep = 0, expd = 0;
                    uint32_t kminA = XR_DIAL_INF, kminB = XR_DIAL_INF;
                    while (bits) {                                    // four distance loads in flight at a time
                        int q[4];
                        uint32_t w[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            q[j] = bits ? __ffsll((long long)bits) - 1 : -1;
                            bits &= bits - 1;                        // (0 stays 0)
                        }
                        int fq[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) fq[j] = (q[j] & 31) * mw + (q[j] < 32 ? wi : wi2);
#pragma unroll
                        for (int j = 0; j < 4; j++) w[j] = q[j] >= 0 ? field[fq[j]] : XR_DIAL_INF;
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            if (q[j] < 0) continue;
                            uint32_t cx = 0, cr, cy = 0, cz = 0;
                            if (XR_DIAL_ASTAR) {
                                xr_divmod((uint32_t)fq[j], uYZ, R.magic_yz, cx, cr);
                                xr_divmod(cr, uZ, R.magic_z, cy, cz);
                            }
                            const uint32_t key = (w[j] >> 2) + heur((int)cx, (int)cy, (int)cz);
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
                    XR_LAP(1);
                    while (expd) {
                        const int qb = __ffsll((long long)expd) - 1;
                        expd &= expd - 1;
                        const int wsel = qb < 32 ? wi : wi2;
                        uint32_t f = (uint32_t)((qb & 31) * mw + wsel);
                        // Run-ahead: a neighbour lowered INTO the current bucket is expanded at once by the same lane (one
                        // successor per node, the others take the open mask), so a distance travels a straight run of
                        // in-bucket nodes within ONE round instead of one round per hop.
                        for (;;) {
                        const uint32_t d4 = field[f] & ~3u;
                        // ---- expand f: planar neighbours along the layer's preferred direction + the two vias
                        uint32_t x, r, y, z;
                        xr_divmod(f, uYZ, R.magic_yz, x, r);
                        xr_divmod(r, uZ, R.magic_z, y, z);
                        const bool vert = (ldir >> z) & 1u;
                        int nf[4];
                        uint32_t len4[4], wn[4], cw[4], old[4];
                        nf[0] = vert ? ((int)y + 1 < Y ? (int)f + Z : -1) : ((int)x + 1 < X ? (int)f + YZ : -1);
                        len4[0] = vert ? el4y(y + 1) : el4x(x + 1);
                        nf[1] = vert ? (y > 0 ? (int)f - Z : -1) : (x > 0 ? (int)f - YZ : -1);
                        len4[1] = vert ? el4y(y) : el4x(x);
                        nf[2] = ((int)z + 1 < Z) ? (int)f + 1 : -1; len4[2] = via4;
                        nf[3] = (z > 0) ? (int)f - 1 : -1;          len4[3] = via4;
#pragma unroll
                        for (int k = 0; k < 4; k++) wn[k] = nf[k] >= 0 ? field[nf[k]] : XR_W_BLOCK;
                        bool refused = false;
                        uint32_t key[4], gcost[4];
                        {   // h of the four neighbours (coordinates differ from f's in one component)
                            const int ix = (int)x, iy = (int)y, iz = (int)z;
                            gcost[0] = vert ? guide_of(ix, iy + 1, iz) : guide_of(ix + 1, iy, iz);
                            gcost[1] = vert ? guide_of(ix, iy - 1, iz) : guide_of(ix - 1, iy, iz);
                            gcost[2] = guide_of(ix, iy, iz + 1); gcost[3] = guide_of(ix, iy, iz - 1);
                            key[0] = vert ? heur(ix, min(iy + 1, Y - 1), iz) : heur(min(ix + 1, X - 1), iy, iz);
                            key[1] = vert ? heur(ix, max(iy - 1, 0), iz) : heur(max(ix - 1, 0), iy, iz);
                            key[2] = heur(ix, iy, min(iz + 1, Z - 1));
                            key[3] = heur(ix, iy, max(iz - 1, 0));
                        }
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const uint32_t cand4 = d4 + len4[k] + ((wn[k] & 2u) ? pen4 : 0u) + gcost[k];
                            cw[k] = cand4 | (wn[k] & 3u);
                            key[k] += cand4 >> 2;                                 // f = d + h
                            // blockage, a distance that does not exist (>= XR_DIST_CAP, spec), or no improvement: nothing to do
                            bool go = wn[k] != XR_W_BLOCK && cand4 < XR_W_USABLE_END && cw[k] < wn[k];
                            if (go && key[k] > best) { refused = true; go = false; }             // bound pruning (on f)
                            if (!go) cw[k] = XR_DIAL_INF;
                        }
                        // the four atomics are issued back to back; their results are looked at afterwards
#pragma unroll
                        for (int k = 0; k < 4; k++) old[k] = cw[k] != XR_DIAL_INF ? atomicMin(&field[nf[k]], cw[k]) : 0u;
                        int next_f = -1;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            if (cw[k] < old[k]) {                      // lowered
                                if (XR_DIAL_CHAIN && next_f < 0 && key[k] < hi) { next_f = nf[k]; continue; }
                                uint32_t oq, orr;                      // the neighbour becomes open
                                xr_divmod((uint32_t)nf[k], umw, magic_mw, oq, orr);
                                atomicOr(&s_open[orr], 1u << oq);
                                atomicMin(&s_wmin[orr], key[k]);
                                lmin = key[k] < lmin ? key[k] : lmin;
                            }
                        }
                        if (refused) xr_mask_or(s_defer, f, umw, magic_mw);
                        if (next_f < 0) break;
                        f = (uint32_t)next_f;
                        }
                    }
                    XR_LAP(2);
                }
                lmin = wmA < lmin ? wmA : lmin;
                lmin = wmB < lmin ? wmB : lmin;
            }
#endif
            XR_LAP(1);
            lmin = xr_wave_min_u32(lmin);
            if ((tid & 63) == 0 && lmin != XR_DIAL_INF) atomicMin(&s_min[nx1], lmin);
            nrounds++;
#if defined(XR_PHASE_TIMING) && !defined(XR_COUNT_HOPS) && !defined(XR_PROBE_SETUP)
            if (tid == XR_TIMING_TID) _ph[7] += 1;
#endif
            __syncthreads();
            XR_LAP(6);
            cur = nx1;
        }
        XR_LAP(6);

        if (aborted) {                        // uniform: the remaining pins are charged as unreachable, nothing is traced
            if (tid == 0) { d_vio += s_remaining; status |= XR_ENV_ROUTER_ABORT | XR_ENV_UNREACHABLE; s_remaining = 0; s_target_i = -1; }
        } else
        // ---- nearest access point of an unconnected pin; ties -> lowest flat index (wave 0) --------
        if (tid < 64) {
            // (distance, flat index) of the best target: one LDS 64-bit atomic min per candidate lane
            for (int i = tid; i < nap; i += 64) {
                if (s_ap_conn[i]) continue;
                const uint32_t w = field[s_ap_f[i]];
                if (w >= XR_W_UNREACHED) continue;
                atomicMin(&s_tkey, ((unsigned long long)(w >> 2) << 32) | (unsigned)s_ap_f[i]);
            }
            __builtin_amdgcn_wave_barrier();
            const unsigned long long bestk = s_tkey;          // (same wave: LDS operations complete in order)
            int best_i = -1;
            if (bestk != ~0ULL) {                 // AP slot holding that node (node ids are unique per net)
                const int bf = (int)(bestk & 0xFFFFFFFFu);
                for (int i0 = 0; i0 < nap && best_i < 0; i0 += 64) {
                    const int i = i0 + tid;
                    const unsigned long long mm = __ballot(i < nap && (int)s_ap_f[i] == bf);
                    if (mm) best_i = i0 + __ffsll((long long)mm) - 1;
                }
            }
            if (tid == 0) s_target_i = best_i;

            if (best_i < 0) {                     // every remaining pin unreachable
                if (tid == 0) {
                    d_vio += s_remaining;
                    status |= XR_ENV_UNREACHABLE;
                    s_remaining = 0;
                }
            } else {
                // ---- deterministic back-trace: first predecessor in the order E,S,W,N,U,D (the reference's own
                // direction order, build_3Dgrid.py:127).  TWO hops per LDS round trip: lanes 0..5 test one direction of the
                // current node v each, lanes 6..41 test — speculatively, for each of the six possible predecessors u1 of v — one
                // direction of u1 each (their addresses depend on v's coordinates only, so both levels load at once).  ballot +
                // ffs picks the first match of each level (wave-uniform lane ids: the winners' values come through v_readlane).
                // The field is only READ here; claimed nodes are zeroed afterwards.
                int v = __builtin_amdgcn_readfirstlane((int)s_ap_f[best_i]);
                uint32_t vw = field[v];
                uint32_t ux, ur, uy, uz;
                xr_divmod((uint32_t)v, uYZ, R.magic_yz, ux, ur);
                xr_divmod(ur, uZ, R.magic_z, uy, uz);
                int x = (int)ux, y = (int)uy, z = (int)uz;
                const int lvl2 = tid >= 6 && tid < 42;
                const int d1 = tid < 6 ? tid : (tid < 42 ? (tid - 6) / 6 : 6);         // direction of the first hop this lane looks at
                const int d2 = lvl2 ? (tid - 6) % 6 : 6;                               // ... and of the second (level-2 lanes)
                // candidate predecessor of node (f; cx, cy, cz) in direction d (0..5 = E,S,W,N,U,D): flat index or -1, edge length x4
                auto pred_of = [&](int f, int cx, int cy, int cz, int d, uint32_t& len4) -> int {
                    // (no switch: six divergent cases would run one after the other, each with its own LDS round trip)
                    const bool vt = (ldir >> cz) & 1u;
                    const int ddx = (d == 0) - (d == 2), ddy = (d == 3) - (d == 1), ddz = (d == 4) - (d == 5);
                    const int nx = cx + ddx, ny = cy + ddy, nz = cz + ddz;
                    const bool along = ddz != 0 || (ddx != 0 && !vt) || (ddy != 0 && vt);      // planar moves follow the layer's direction
                    const bool inb = d < 6 && along && (unsigned)nx < (unsigned)X && (unsigned)ny < (unsigned)Y && (unsigned)nz < (unsigned)Z;
                    const uint32_t* ctab = ddx ? s_xc : s_yc;
                    const int c0 = ddx ? cx : cy, c1 = c0 + ddx + ddy;
                    const uint32_t ca = ctab[c0 + 1], cb = ctab[(ddz ? c0 : c1) + 1];
                    len4 = ddz ? via4 : (cb > ca ? cb - ca : ca - cb);
                    return inb ? f + ddx * YZ + ddy * Z + ddz : -1;
                };
                auto claim_node = [&](int node, uint32_t w, int src, uint32_t pl4) {     // thread 0: node joins the path, left through `src`
                    uint32_t cq, cr;
                    xr_divmod((uint32_t)node, umw, magic_mw, cq, cr);
                    if (w & 2u) { d_vio += 1; d_held += 1; }
                    atomicOr(&s_claim[cr], 1u << cq);
                    if (plen < b.path_cap) path[plen] = node;
                    plen++;
                    fnv_mix(h, (uint32_t)node);
                    if (src >= 4) d_via += 1; else d_wl += (int)(pl4 >> 2);
                };
                for (int nt = 0; (vw >> 2) > 0; nt++) {
                    if (nt > N) { if (tid == 0) status |= 0x100; break; }    // (distances strictly decrease: a path is simple; hang guard)
                    const uint32_t need4 = (vw & ~3u) - ((vw & 2u) ? pen4 : 0u) - guide_of(x, y, z);   // pred distance + edge, x4
                    uint32_t len1 = 0, len2 = 0;
                    const int u1 = pred_of(v, x, y, z, d1, len1);
                    const int x1 = x + (d1 == 0) - (d1 == 2), y1 = y + (d1 == 3) - (d1 == 1), z1 = z + (d1 == 4) - (d1 == 5);
                    const int u2 = (lvl2 && u1 >= 0) ? pred_of(u1, x1, y1, z1, d2, len2) : -1;
                    // (both loads unconditionally, from a safe address where there is no candidate: no branch, one round trip)
                    const uint32_t rw1 = field[u1 >= 0 ? u1 : v], rw2 = field[u2 >= 0 ? u2 : v];
                    const uint32_t uw1 = u1 >= 0 ? rw1 : XR_W_BLOCK, uw2 = u2 >= 0 ? rw2 : XR_W_BLOCK;
                    const bool use1 = (uw1 - 1u) < (XR_W_USABLE_END - 1u);
                    const bool ok1 = tid < 6 && use1 && (uw1 & ~3u) + len1 == need4;
                    const bool ok2 = use1 && (uw2 - 1u) < (XR_W_USABLE_END - 1u) &&
                                     (uw2 & ~3u) + len2 == (uw1 & ~3u) - ((uw1 & 2u) ? pen4 : 0u) - guide_of(x1, y1, z1);
                    const unsigned long long mm = __ballot(ok1), mm2 = __ballot(ok2);
                    if (mm == 0) { if (tid == 0) status |= 0x100; break; }     // inconsistent field: cannot happen
                    const int src = __ffsll((long long)mm) - 1;                // wave-uniform
                    const int pu = __builtin_amdgcn_readlane(u1, src);
                    const uint32_t puw = (uint32_t)__builtin_amdgcn_readlane((int)uw1, src);
                    const uint32_t pl4 = (uint32_t)__builtin_amdgcn_readlane((int)len1, src);
                    if (tid == 0) claim_node(v, vw, src, pl4);
                    x += (src == 0) - (src == 2);
                    y += (src == 3) - (src == 1);
                    z += (src == 4) - (src == 5);
                    v = pu; vw = puw;
                    if ((vw >> 2) == 0) break;                                 // a source: the component is reached
                    // ---- the second hop, already looked at by lanes 6 + 6 src .. 6 + 6 src + 5
                    const uint32_t m6 = (uint32_t)(mm2 >> (6 + 6 * src)) & 63u;
                    if (m6 == 0) { if (tid == 0) status |= 0x100; break; }
                    const int srcb = __ffs((int)m6) - 1;
                    const int lane2 = 6 + 6 * src + srcb;
                    const int pu2 = __builtin_amdgcn_readlane(u2, lane2);
                    const uint32_t puw2 = (uint32_t)__builtin_amdgcn_readlane((int)uw2, lane2);
                    const uint32_t pl42 = (uint32_t)__builtin_amdgcn_readlane((int)len2, lane2);
                    if (tid == 0) claim_node(v, vw, srcb, pl42);
                    x += (srcb == 0) - (srcb == 2);
                    y += (srcb == 3) - (srcb == 1);
                    z += (srcb == 4) - (srcb == 5);
                    v = pu2; vw = puw2;
                }
                if (tid == 0 && (status & 0x100)) {
                    s_remaining = 0;              // never taken on a consistent field; avoids spinning
                } else if (tid == 0) {
                    // terminal node of the component: claimed (and recorded) only if nobody holds it yet
                    if (owner[v] == 0) {
                        owner[v] = claim_val;
                        if (plen < b.path_cap) path[plen] = v;
                        plen++;
                        fnv_mix(h, (uint32_t)v);
                    }
                    s_remaining -= 1;
                }
            }
        }
        __syncthreads();
        XR_LAP(3);
        // path nodes and the reached pin's access points become sources of the next search
        {
            const int ti = s_target_i;
            if (ti >= 0) {
                const short pin = s_ap_pin[ti];
                for (int i = tid; i < nap; i += nthr)
                    if (s_ap_pin[i] == pin) { s_ap_conn[i] = 1; make_source(s_ap_f[i]); }
                for (int wi = tid; wi < mw; wi += nthr) {
                    uint32_t m = s_claim[wi];
                    if (m) {
                        s_claim[wi] = 0;
                        atomicOr(&s_open[wi], m);
                        s_wmin[wi] = 0u;
                        while (m) {
                            const int f = (__ffs((int)m) - 1) * mw + wi;
                            m &= m - 1;
                            field[f] &= 3u;
                            // claim the path node if nobody holds it (by many threads at once, instead of one dependent
                            // HBM load per node inside the serial back-trace)
                            if (owner[f] == 0) owner[f] = claim_val;
                        }
                    }
                }
            }
        }
        XR_LAP(4);
        // (the barrier at the top of the loop orders these writes before the next search / the exit test)
    }
    if (!XR3_ALL_ATTEMPTS || !V2 || b.maze_end_iter <= 1) break;
    // ---- XR-Maze v2: does the attempt stand?  Its path uses a node held by another net and attempts are left: rip it up ----
    if (tid == 0) s_retry = (d_held > 0 && attempt + 1 < b.maze_end_iter) ? 1 : 0;
    __syncthreads();
    const bool retry = s_retry != 0;
    for (int f = tid; f < N; f += nthr)                       // tentative claims: accepted (-a -> a) or undone (-a -> 0)
        if (owner[f] == (int16_t)-a) owner[f] = retry ? (int16_t)0 : (int16_t)a;
    if (!retry) break;
    attempt++;
    pen4 = ((uint32_t)b.pen_cost << 2) << attempt;
    if (tid == 0) { d_vio = 0; d_wl = 0; d_via = 0; plen = 0; d_held = 0; status = XR_ENV_OK; h = h0; }
    __syncthreads();                                          // the owner grid is clean again before the field is rebuilt
    build_field();
    __syncthreads();
    for (int i = tid; i < nap; i += nthr) {
        const bool iso = s_ap_conn[i] == 2;
        const bool first = s_ap_pin[i] == (short)s_first_pin;
        s_ap_conn[i] = iso ? 2 : (first ? 1 : 0);
        if (first && !iso) make_source(s_ap_f[i]);
    }
    if (tid == 0) s_remaining = s_npins - 1 - n_isolated;
    // (the barrier at the top of the search loop orders all of this)
    }

    if (tid == 0) {
        if (n_isolated > 0) { d_vio += n_isolated; status |= XR_ENV_UNREACHABLE; }
        xr_step_epilogue(b, e, a, d_vio, d_wl, d_via, plen, status, nrounds, h);
    }
    XR_LAP(5);
    XR_TDUMP();
}

// ------------------------------------------------------------------------------------------------
// HBM-scratch form for regions whose field does not fit LDS (BASELINE config 5: 256x256x12 = 786 k nodes).
//
// Same algorithm, different placement — and NO per-step sweep over the N nodes: the per-env field lives in HBM scratch in
// a persistent CLEAN state (0xFFFFFFFE = "never touched in this route"); a node's word is created the first time a
// neighbour relaxes it (blockage / held flags derived from node_net / owner on the spot), every first touch is logged
// in a `touched` list and the route ends by resetting exactly those nodes.  A route that explores 29 k of 786 k nodes
// (the BASELINE config 5 average) therefore moves ~1 MB instead of the 6.3 MB a grid build alone would cost.
//
//   field  u32[n_max]   CLEAN | (distance << 2) | (held << 1) | 1           (all accesses: L2-scope atomics / sc1 loads)
//   open   u32[n_max/32]         node bit f & 31 of word f >> 5 (flat order: a word = 32 consecutive nodes)
//   defer  a LIST of nodes (an expanded node with an edge refused by the bound), re-opened when the next search starts
//   wmin   u32[n_max/32]         lower bound of the open distances of a word        (HBM)
//   gmin   u32[n_max/1024]       lower bound per group of 32 words                  (LDS)
//
// A round is a staged pipeline over compact work lists in LDS, so that every global round trip is taken by ALL work items
// at once (the latency of an L2 atomic is ~1-2 us; a lane-per-word scan as in the LDS form would serialise them):
//   A  groups with gmin < hi -> list G          A2  their words with wmin < hi -> list A
//   B  take the open bits of the words of A -> node list N
//   C  classify the nodes of N by distance: beyond the bucket -> back into open / wmin / gmin; inside -> list E (f, d)
//   D  one lane per (node of E, direction): relax the neighbour.  A neighbour lowered INTO the bucket goes to the next E list
//      and stage D runs again (run-ahead within the round: one barrier + two round trips per hop instead of a full round).
// Lists have fixed capacities; whatever does not fit stays in (or is put back into) the open mask and is picked up by
// the next chunk / round — capacity never affects the result.
// ------------------------------------------------------------------------------------------------
#define XR_BIG_CLEAN 0xFFFFFFFEu
// list capacities XR_BIG_CA / _CN / _CE / _MAXG: xr_device.h (the host sizes the LDS from them)

__device__ __forceinline__ uint32_t xr_ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void xr_st(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// V2: XR-Maze v2 knobs compiled in (guide cost, rip-up-and-reroute), as in the LDS form
template <bool V2>
__device__ __forceinline__ void xr_dial_route_env_big(const XrBatchDev& b, const int e, const int a, char* smem) {
    __shared__ int s_ap_f[XR_MAX_AP_PER_NET];
    __shared__ short s_ap_pin[XR_MAX_AP_PER_NET];
    __shared__ unsigned char s_ap_conn[XR_MAX_AP_PER_NET];
    __shared__ uint32_t s_min[3], s_bst[3];
    __shared__ unsigned long long s_tkey;
    __shared__ int s_hbk[XR_HB_MAX][6];                         // heuristic: one box per unconnected pin (x, y: coordinates x4; z), see the search start
    __shared__ short s_hbpin[XR_HB_MAX];
    __shared__ int s_nhb;
    __shared__ int s_remaining, s_target_i, s_first_pin, s_npins, s_niso, s_src_iso;
    __shared__ int s_pocket[4][XR_POCKET_CAP + 8];
    __shared__ int s_nG, s_nN, s_nE[3], s_ntouched, s_plen, s_ndefer;       // s_nE: three counters taking turns over the two E lists (one barrier per run-ahead pass)
    __shared__ int s_gb[4], s_retry, s_ngb;                   // XR-Maze v2: bounding box of the net's access points (track indices), rip-up decision
    __shared__ int4 s_gbx[XR_GUIDE_MAX_BOXES];                // ... and its guide (xr_guide_load)

    const int tid = threadIdx.x;
    const int nthr = blockDim.x;
    if (!xr_step_prologue(b, e, a)) return;

    XR_T0();
    const XrRegionDev R = b.regions[b.env_region[e]];
    const int X = R.X, Y = R.Y, Z = R.Z, N = R.N;
    const int YZ = Y * Z;
    const uint32_t ldir = R.ldir_mask;
    const int mw = (N + 31) >> 5;
    const int ng = (mw + 31) >> 5;
    const int round_cap = b.round_cap > 0 ? b.round_cap : 1024 + N;
    const int16_t* __restrict__ node_net = b.rg_node_net + R.node_off;
    int16_t* __restrict__ owner = b.owner + (int64_t)e * b.n_max;
    const int64_t mwg = (int64_t)(b.n_max >> 5) + 1;
    uint32_t* __restrict__ fieldg = b.dg_field + (int64_t)e * b.n_max;
    uint32_t* __restrict__ openg = b.dg_masks + (int64_t)e * 2 * mwg;
    uint32_t* __restrict__ wming = openg + mwg;
    uint32_t* __restrict__ touchg = b.dg_touch + (int64_t)e * b.n_max;
    uint32_t* __restrict__ pathg = b.dg_path + (int64_t)e * b.n_max * 2;      // [0, n_max): path of the current trace (a path is simple)
    const int defer_cap = b.n_max;
    uint32_t* __restrict__ deferl = pathg + b.n_max;                          // [n_max, 2 n_max): deferred nodes

    // LDS carve: gmin u32[MAXG] | G u16[MAXG] | A u32[CA] | N u32[CN] | E0, E1 uint2[CE] | el4x | el4y
    uint32_t* s_gmin = reinterpret_cast<uint32_t*>(smem);
    unsigned short* s_G = reinterpret_cast<unsigned short*>(s_gmin + XR_BIG_MAXG);
    uint32_t* s_A = reinterpret_cast<uint32_t*>(s_G + XR_BIG_MAXG);
    uint32_t* s_N = s_A + XR_BIG_CA;
    uint2* s_E0 = reinterpret_cast<uint2*>(s_N + XR_BIG_CN);
    uint2* s_E1 = s_E0 + XR_BIG_CE;
    uint32_t* s_xc = reinterpret_cast<uint32_t*>(s_E1 + XR_BIG_CE);          // coordinate tables, see the LDS form
    uint32_t* s_yc = s_xc + (b.x_max + 2);

    for (int i = tid; i < ng; i += nthr) s_gmin[i] = XR_DIAL_INF;
    for (int i = tid; i <= X + 1; i += nthr)
        s_xc[i] = (uint32_t)(b.coords[R.xs_off + min(max(i - 1, 0), X - 1)] - b.coords[R.xs_off]) << 2;
    for (int i = tid; i <= Y + 1; i += nthr)
        s_yc[i] = (uint32_t)(b.coords[R.ys_off + min(max(i - 1, 0), Y - 1)] - b.coords[R.ys_off]) << 2;
    const int ap_lo = b.net_csr[R.net_off + a], ap_hi = b.net_csr[R.net_off + a + 1];
    const int nap = ap_hi - ap_lo;
    if (tid == 0) {
        s_first_pin = 0x7FFFFFFF; s_npins = 0; s_ntouched = 0; s_ndefer = 0; s_niso = 0; s_src_iso = 0;
        s_gb[0] = 0x7FFFFFFF; s_gb[1] = -1; s_gb[2] = 0x7FFFFFFF; s_gb[3] = -1;
    }
    __syncthreads();
    for (int i = tid; i < nap; i += nthr) {
        const int pin = b.ap_pin[R.ap_off + ap_lo + i];
        s_ap_f[i] = b.ap_node[R.ap_off + ap_lo + i];
        s_ap_pin[i] = (short)pin;
        s_ap_conn[i] = 0;
        atomicMin(&s_first_pin, pin);
        if (V2 && b.guide_cost) {                // XR-Maze v2: the net's guide = bounding box of all its access points (+ margin)
            const int apf = s_ap_f[i];
            const int gy = (apf / Z) % Y, gx = apf / YZ;
            atomicMin(&s_gb[0], gx); atomicMax(&s_gb[1], gx); atomicMin(&s_gb[2], gy); atomicMax(&s_gb[3], gy);
        }
    }
    __syncthreads();
    // flags of a node for THIS net: 0 = blockage, else 1 | held << 1
    auto node_flags = [&](int f) -> uint32_t {
        const int nn = node_net[f], ow = owner[f];
        if (nn == -1) return 0u;
        return 1u | (((ow != 0 && ow != a && !(V2 && ow == -a)) || (nn > 0 && nn != a)) ? 2u : 0u);     // (-a: a tentative claim of this very route)
    };
    // first touch of a node in this route: remember it for the final reset
    auto touch = [&](uint32_t f) { const int k = atomicAdd(&s_ntouched, 1); touchg[k] = f; };
    // a node becomes a source: distance 0, open
    auto make_source = [&](int f) {
        const uint32_t old = xr_ld(&fieldg[f]);
        if (old == XR_BIG_CLEAN) touch((uint32_t)f);
        xr_st(&fieldg[f], old == XR_BIG_CLEAN ? node_flags(f) : (old & 3u));
        atomicOr(&openg[f >> 5], 1u << (f & 31));
        xr_st(&wming[f >> 5], 0u);
        s_gmin[f >> 10] = 0u;
    };
    for (int i = tid; i < nap; i += nthr) {
        const short pin = s_ap_pin[i];
        bool seen = false;
        for (int j = 0; j < i; j++) seen |= (s_ap_pin[j] == pin);
        if (!seen) atomicAdd(&s_npins, 1);
        unsigned char cflag = seen ? 0 : 0x80;            // bit 7: first access point of its pin (for xr_mark_isolated_pins)
        if (pin == (short)s_first_pin) { cflag |= 1; make_source(s_ap_f[i]); }
        s_ap_conn[i] = cflag;
    }
    __syncthreads();
    // pins in closed pockets are never searched for (see xr_mark_isolated_pins)
    xr_mark_isolated_pins(nap, s_ap_f, s_ap_pin, s_ap_conn, X, Y, Z, ldir, R.magic_yz, R.magic_z,
                          [&](int f) { return node_net[f] == -1; }, s_pocket, &s_niso, &s_src_iso, s_first_pin, XR_POCKET_BUDGET_BIG);
    __syncthreads();
    for (int i = tid; i < nap; i += nthr) s_ap_conn[i] &= 0x7F;
    const int n_isolated = s_src_iso ? s_npins - 1 : s_niso;       // unreachable pins known up front
    if (tid == 0) s_remaining = s_npins - 1 - n_isolated;
    XR_LAP(0);

    const uint32_t via4 = (uint32_t)b.via_cost << 2;
    uint32_t pen4 = (uint32_t)b.pen_cost << 2;              // (XR-Maze v2: doubled by every rip-up-and-reroute attempt)
    const uint32_t delta = R.w_min * (uint32_t)b.dial_mult_big;
    const uint32_t uYZ = (uint32_t)YZ, uZ = (uint32_t)Z;
    int d_vio = 0, d_wl = 0, d_via = 0, plen = 0, status = XR_ENV_OK, d_held = 0;   // thread 0 only
    int nrounds = 0;
    const uint64_t h0 = (tid == 0) ? b.hash[e] : 0;
    uint64_t h = h0;
    int32_t* __restrict__ path = b.path + (int64_t)e * b.path_cap;

    // put a node (back) into the open structure with distance d
    auto open_insert = [&](uint32_t f, uint32_t d) {
        atomicOr(&openg[f >> 5], 1u << (f & 31));
        atomicMin(&wming[f >> 5], d);
        atomicMin(&s_gmin[f >> 10], d);
    };

    // XR-Maze v2, neutral by default: guide cost outside the net's guide box; tentative claims (owner = -a) until
    // the attempt stands
    const uint32_t guide4 = V2 ? (uint32_t)b.guide_cost << 2 : 0u;
    if (V2 && b.guide_cost) {
        xr_guide_load(b, R, a, s_gb, Z, s_gbx, &s_ngb, tid);
        __syncthreads();
    }
    const int ngb = (V2 && b.guide_cost) ? s_ngb : 0;
    int4 gb0 = make_int4(0, 0, 0, 0);
    if (V2 && b.guide_cost) {
        const int4 t = s_gbx[0];
        gb0 = make_int4(__builtin_amdgcn_readfirstlane(t.x), __builtin_amdgcn_readfirstlane(t.y), __builtin_amdgcn_readfirstlane(t.z), 0);
    }
    auto guide_of = [&](int x, int y, int z) -> uint32_t {
        if (!V2) return 0u;
        return (guide4 != 0u && !xr_guide_has(s_gbx, ngb, gb0, x, y, z)) ? guide4 : 0u;
    };
    const int16_t claim_val = (int16_t)(XR3_ALL_ATTEMPTS && V2 && b.maze_end_iter > 1 ? -a : a);      // (one attempt: its claims are final)
#ifdef XR3_V2_ALL_ATTEMPTS
    int attempt = 0;
#else           // the rip-up-and-reroute loop has one possible outcome: the last attempt's route (xr_dial3.h)
    int attempt = V2 ? max(b.maze_end_iter, 1) - 1 : 0;
    if (V2) pen4 <<= attempt;
#endif

    for (;;) {                                              // attempts (exactly one unless maze_end_iter > 1)
    for (;;) {
        // ---- new search ---------------------------------------------------------------------------------
        if (tid == 0) {
            s_min[0] = 0; s_min[1] = XR_DIAL_INF; s_min[2] = XR_DIAL_INF;
            s_bst[0] = XR_DIAL_INF; s_bst[1] = XR_DIAL_INF; s_bst[2] = XR_DIAL_INF;
            s_tkey = ~0ULL;
        }
        // deferred nodes are looked at again (the list overflowed: every reached node is — re-expanding one is harmless)
        {
            const int ndf = s_ndefer;
            const bool ovf = ndf > defer_cap;
            const int cnt = ovf ? s_ntouched : ndf;
            const uint32_t* __restrict__ src = ovf ? touchg : deferl;
            for (int i = tid; i < cnt; i += nthr) {
                const uint32_t f = src[i];
                const uint32_t w = xr_ld(&fieldg[f]);
                if (w < XR_W_USABLE_END) open_insert(f, w >> 2);
            }
        }
        __syncthreads();
        if (tid == 0) s_ndefer = 0;
        __syncthreads();
        if (s_remaining <= 0) break;          // uniform
        // heuristic of this search (round 4): h(v) = min over the unconnected (and not isolated) PINS of the distance from v to the box of
        // that pin's access points (coordinate differences + one via cost per layer).  A minimum of consistent lower bounds is one, and
        // it is 0 on every target, so the results are the oracle's as before — but where ONE box around all unconnected pins is 0
        // everywhere between them (a 3-4 pin net spread over 70 x 70 tracks flooded its whole box: 40-50 k touched nodes, the routes
        // that decide a launch, profiles/r04_b_config5_route_distribution.txt), this one leads the search to the nearest pin.
        // The first XR_HB_MAX - 1 pins get a box each, any further pins share the last one (still a lower bound).
        if (tid == 0) {
            int nh = 0;
            for (int i = 0; i < nap; i++) {
                if (s_ap_conn[i]) continue;
                uint32_t ax, ar, ay, az;
                xr_divmod((uint32_t)s_ap_f[i], uYZ, R.magic_yz, ax, ar);
                xr_divmod(ar, uZ, R.magic_z, ay, az);
                const int cx = (int)s_xc[ax + 1], cy = (int)s_yc[ay + 1], cz = (int)az;
                const short pin = s_ap_pin[i];
                int k = 0;
                while (k < nh && s_hbpin[k] != pin) k++;
                if (k == nh) {
                    if (nh < XR_HB_MAX) {
                        s_hbpin[nh] = pin;
                        s_hbk[nh][0] = cx; s_hbk[nh][1] = cx; s_hbk[nh][2] = cy; s_hbk[nh][3] = cy; s_hbk[nh][4] = cz; s_hbk[nh][5] = cz;
                        nh++;
                        continue;
                    }
                    k = XR_HB_MAX - 1;
                }
                s_hbk[k][0] = min(s_hbk[k][0], cx); s_hbk[k][1] = max(s_hbk[k][1], cx);
                s_hbk[k][2] = min(s_hbk[k][2], cy); s_hbk[k][3] = max(s_hbk[k][3], cy);
                s_hbk[k][4] = min(s_hbk[k][4], cz); s_hbk[k][5] = max(s_hbk[k][5], cz);
            }
            s_nhb = nh;
        }
        __syncthreads();
        const int nhb = s_nhb;
        auto heur_c = [&](int xc, int yc, int z) -> uint32_t {                 // (from coordinates x4)
            uint32_t hmin = 0xFFFFFFFFu;
            for (int k = 0; k < nhb; k++) {
                const int hx = max(0, max(s_hbk[k][0] - xc, xc - s_hbk[k][1])), hy = max(0, max(s_hbk[k][2] - yc, yc - s_hbk[k][3]));
                const int hz = max(0, max(s_hbk[k][4] - z, z - s_hbk[k][5]));
                const uint32_t hv = ((uint32_t)(hx + hy) >> 2) + (uint32_t)hz * (uint32_t)b.via_cost;
                hmin = hv < hmin ? hv : hmin;
            }
            return hmin;
        };
        auto heur = [&](int x, int y, int z) -> uint32_t { return heur_c((int)s_xc[x + 1], (int)s_yc[y + 1], z); };
        int cur = 0;
        bool aborted = false;                 // (round cap: see the LDS form)
        for (int nsr = 0;; nsr++) {
            const int nx1 = cur == 2 ? 0 : cur + 1, nx2 = nx1 == 2 ? 0 : nx1 + 1;
            const uint32_t m = s_min[cur], best = s_bst[cur];
            if (m == XR_DIAL_INF || m > best) break;                 // uniform
            if (nsr >= round_cap) { aborted = true; break; }         // uniform
            const uint32_t hi = m + delta;
            uint32_t lmin = XR_DIAL_INF;
            if (tid == 0) { s_min[nx2] = XR_DIAL_INF; s_bst[nx2] = XR_DIAL_INF; s_nG = 0; }
            for (int i = tid; i < nap; i += nthr)
                if (!s_ap_conn[i]) { const uint32_t w = xr_ld(&fieldg[s_ap_f[i]]); if (w < XR_W_USABLE_END) atomicMin(&s_bst[nx1], w >> 2); }
            __syncthreads();
            // ---- A: active groups ------------------------------------------------------------------
            for (int g = tid; g < ng; g += nthr) {
                const uint32_t gm = s_gmin[g];
                if (gm < hi) { s_gmin[g] = XR_DIAL_INF; s_G[atomicAdd(&s_nG, 1)] = (unsigned short)g; }
                else lmin = gm < lmin ? gm : lmin;
            }
            __syncthreads();
            const int nG = s_nG;
            // chunks of the active groups: 8 groups (= 256 words = one A list) at a time
            for (int g0 = 0; g0 < nG; g0 += XR_BIG_CA / 32) {
                if (tid == 0) { s_nN = 0; s_nE[0] = 0; s_nE[1] = 0; s_nE[2] = 0; }
                __syncthreads();
                // ---- A2 + B (fused, round 4): active words of these groups, and their open bits -> node list.  The cached minimum is
                // read AND reset by ONE atomic exchange (a word that turns out inactive gets its bound back by an atomicMin — nobody reads
                // it in between: insertions only happen in stages C / D, behind barriers); the exchange has returned, i.e. the reset is
                // done at L2, before the same lane takes the bits — "reset, THEN take the bits, THEN read distances" holds without the
                // barrier and the A list that used to sit between the two stages: two dependent round trips instead of three.
                {
                    constexpr int XR_A2_MAXW = 8;                      // words a thread looks at per chunk (XR_BIG_CA / blockDim.x)
                    uint32_t wds[XR_A2_MAXW], wms[XR_A2_MAXW];
                    int gs[XR_A2_MAXW];
                    for (int base = 0; base < XR_BIG_CA; base += XR_A2_MAXW * nthr) {
#pragma unroll
                        for (int u = 0; u < XR_A2_MAXW; u++) {          // all exchanges of this thread in flight together
                            const int i = base + u * nthr + tid;
                            const int gi = g0 + (i >> 5);
                            wds[u] = 0xFFFFFFFFu; wms[u] = XR_DIAL_INF; gs[u] = 0;
                            if (i < XR_BIG_CA && gi < nG) {
                                const int g = s_G[gi];
                                const int wd = (g << 5) + (i & 31);
                                if (wd < mw) { wds[u] = (uint32_t)wd; gs[u] = g; wms[u] = atomicExch(&wming[wd], XR_DIAL_INF); }
                            }
                        }
                        uint32_t bitsv[XR_A2_MAXW];
#pragma unroll
                        for (int u = 0; u < XR_A2_MAXW; u++) {
                            bitsv[u] = 0u;
                            if (wds[u] == 0xFFFFFFFFu) continue;
                            const uint32_t wm = wms[u];
                            if (wm < hi) bitsv[u] = atomicExch(&openg[wds[u]], 0u);
                            else if (wm != XR_DIAL_INF) { atomicMin(&wming[wds[u]], wm); atomicMin(&s_gmin[gs[u]], wm); lmin = wm < lmin ? wm : lmin; }
                        }
#pragma unroll
                        for (int u = 0; u < XR_A2_MAXW; u++) {
                            uint32_t bits = bitsv[u];
                            const int cnt = __popc(bits);
                            if (!cnt) continue;
                            const uint32_t wd = wds[u];
                            const int pos = atomicAdd(&s_nN, cnt);
                            if (pos + cnt <= XR_BIG_CN) {
                                int k = pos;
                                while (bits) { s_N[k++] = (wd << 5) + (uint32_t)(__ffs((int)bits) - 1); bits &= bits - 1; }
                            } else {                                        // no room in this chunk: back into the mask, seen again next round
                                atomicAdd(&s_nN, -cnt);
                                atomicOr(&openg[wd], bits);
                                atomicMin(&wming[wd], m);
                                atomicMin(&s_gmin[wd >> 5], m);
                                lmin = m < lmin ? m : lmin;
                            }
                        }
                    }
                }
                __syncthreads();
                // ---- C: classify ---------------------------------------------------------------------
                const int nN = min(s_nN, XR_BIG_CN);
                for (int i = tid; i < nN; i += nthr) {
                    const uint32_t f = s_N[i];
                    const uint32_t w = xr_ld(&fieldg[f]);
                    uint32_t cx, cr, cy, cz;
                    xr_divmod(f, uYZ, R.magic_yz, cx, cr);
                    xr_divmod(cr, uZ, R.magic_z, cy, cz);
                    const uint32_t d = (w >> 2) + heur((int)cx, (int)cy, (int)cz);          // the key f = d + h
                    bool keep = d >= hi;
                    if (!keep) {
                        const int pos = atomicAdd(&s_nE[0], 1);
                        if (pos < XR_BIG_CE) s_E0[pos] = make_uint2(f, w & ~3u);
                        else keep = true;                               // no room: stays open
                    }
                    if (keep) { open_insert(f, d); lmin = d < lmin ? d : lmin; }
                }
                __syncthreads();
                // ---- D: relax, one lane per (node, direction); run ahead inside the bucket -----------------------
                // (round 5) the counters rotate over THREE slots — this pass reads ec, appends to ec1, and clears ec2 for the pass after next —
                // so a pass ends with ONE workgroup barrier (round 4: barrier, thread 0 clears the consumed counter, barrier); the two E lists
                // still take turns (the list pass k + 1 appends to is the one pass k read)
                int eb = 0, ec = 0;
                for (int npass = 0;; npass++) {
                    const int nE = min(s_nE[ec], XR_BIG_CE);
                    if (nE == 0) break;                                  // uniform
                    const int ec1 = ec == 2 ? 0 : ec + 1, ec2 = ec1 == 2 ? 0 : ec1 + 1;
                    if (tid == 0) s_nE[ec2] = 0;
                    if (npass >= round_cap) {            // hang guard (every pass lowers field words: finite anyway): what is left stays open
                        uint2* El = eb ? s_E1 : s_E0;
                        for (int it = tid; it < nE; it += nthr) { open_insert(El[it].x, m); }
                        lmin = m < lmin ? m : lmin;
                        __syncthreads();
                        if (tid == 0) s_nE[ec] = 0;
                        __syncthreads();
                        break;
                    }
                    uint2* Ecur = eb ? s_E1 : s_E0;
                    uint2* Enxt = eb ? s_E0 : s_E1;
#ifdef XR_BIG_PASS_PROBE       // (tools/config5_pass_split.py: where a run-ahead pass spends its time, as thread 0 sees it)
                    const long long pp0 = clock64();
                    long long pp1 = 0, pp2 = 0, pp3 = 0;
#endif
                    // (Round 5, measured and NOT adopted — both bit-exact: look-ahead loads of the next pass's words by the idle half of the
                    //  workgroup, +12 % (profiles/r05_g_*); "flags forward": the pass that lowers a node fetches the flags of ITS neighbours beside
                    //  its atomic and hands them on in the E entry, so the next pass goes straight to its atomics — one L2 round trip per pass
                    //  instead of two, yet +17 % (profiles/r05_j_*, tools/archive/xr_big_flags_forward.patch): the pass-split probe shows why —
                    //  ~1.1 k of a pass's ~3.5 k cycles are the index / heuristic arithmetic in front of the atomic, the word load hides
                    //  behind it, and six more loads per lane queue up behind the atomic whose result the pass waits for.)
                    for (int it = tid; it < 4 * nE; it += nthr) {
                        const uint2 en = Ecur[it >> 2];
                        const int dir = it & 3;
                        const uint32_t f = en.x, d4 = en.y;
                        uint32_t x, r, y, z;
                        xr_divmod(f, uYZ, R.magic_yz, x, r);
                        xr_divmod(r, uZ, R.magic_z, y, z);
                        const bool vert = (ldir >> z) & 1u;
                        // (no branch per direction — the wave would run the four arms one after the other, each with its own round
                        //  trips; loads from safe addresses, all issued before the first use)
                        const int sgn = (dir & 1) ? -1 : 1;
                        const bool planar = dir < 2;
                        const int ddx = (planar && !vert) ? sgn : 0, ddy = (planar && vert) ? sgn : 0, ddz = planar ? 0 : sgn;
                        const int nx = (int)x + ddx, ny = (int)y + ddy, nz = (int)z + ddz;
                        const bool inb = (unsigned)nx < (unsigned)X && (unsigned)ny < (unsigned)Y && (unsigned)nz < (unsigned)Z;
                        const int nf = inb ? (int)f + ddx * YZ + ddy * Z + ddz : (int)f;
                        const uint32_t wn = xr_ld(&fieldg[nf]);
#if XR_BIG_SPEC_FLAGS
                        const int nn0 = node_net[nf], ow0 = owner[nf];        // flags of a first touch, loaded with the word, not after it
#endif
                        const uint32_t xq = s_xc[x + 1], yq = s_yc[y + 1];
                        const uint32_t cb = (vert ? s_yc : s_xc)[(vert ? (int)y : (int)x) + 1 + (planar ? sgn : 0)];
                        const uint32_t ca = vert ? yq : xq;
                        const uint32_t len4 = planar ? (sgn > 0 ? cb - ca : ca - cb) : via4;
                        // (round 5) what does NOT depend on the words in flight — the neighbour's heuristic (a loop over the pin boxes in LDS), its
                        // guide cost — is computed here, in the shadow of the loads; it used to sit between the load and the atomic, on the chain
                        // (tools/config5_pass_split.py: 1.5 k cycles from "loads returned" to "atomic returned" for an L2 atomic of ~1 k)
                        const uint32_t hq = heur_c((int)((planar && !vert) ? cb : xq), (int)((planar && vert) ? cb : yq), nz);
                        const uint32_t gq = guide_of(nx, ny, nz);
#ifdef XR_BIG_PASS_PROBE
                        if (tid == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); pp1 = clock64(); }
#endif
                        if (!inb) continue;
                        uint32_t fl;
                        if (wn == XR_BIG_CLEAN) {                                 // first touch: derive the flags
#if XR_BIG_SPEC_FLAGS
                            fl = nn0 == -1 ? 0u : (1u | (((ow0 != 0 && ow0 != a && !(V2 && ow0 == -a)) || (nn0 > 0 && nn0 != a)) ? 2u : 0u));
#else
                            fl = node_flags(nf);
#endif
                            if (fl == 0u) continue;
                        } else fl = wn & 3u;
                        const uint32_t cand4 = d4 + len4 + ((fl & 2u) ? pen4 : 0u) + gq;
                        if (cand4 >= XR_W_USABLE_END) continue;
                        const uint32_t cw = cand4 | fl;
                        if (cw >= wn) continue;
                        const uint32_t key = (cand4 >> 2) + hq;   // f = d + h
                        if (key > best) {                              // bound pruning: f is looked at again by the next search
                            const int k = atomicAdd(&s_ndefer, 1);
                            if (k < defer_cap) deferl[k] = f;
                            continue;
                        }
                        const uint32_t old = atomicMin(&fieldg[nf], cw);
#ifdef XR_BIG_PASS_PROBE
                        if (tid == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); pp2 = clock64(); }
#endif
                        // (round 5) both list positions are asked for before either is used: two LDS atomics in flight together instead of
                        // one after the other (the touched-list slot, then the E-list slot: 650 cycles behind the atomic, tools/config5_pass_split.py)
                        const bool first = old == XR_BIG_CLEAN, chain_try = cw < old && key < hi;
                        int tk = 0, pos = XR_BIG_CE;
                        if (first) tk = atomicAdd(&s_ntouched, 1);
                        if (chain_try) pos = atomicAdd(&s_nE[ec1], 1);
                        if (first) touchg[tk] = (uint32_t)nf;
                        if (cw < old) {
                            const bool chained = pos < XR_BIG_CE;
                            if (chained) Enxt[pos] = make_uint2((uint32_t)nf, cand4);
                            else { open_insert((uint32_t)nf, key); lmin = key < lmin ? key : lmin; }
                        }
#ifdef XR_BIG_PASS_PROBE
                        if (tid == 0 && it == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); pp3 = clock64(); }
#endif
                    }
                    // (LDS-only barrier, round 4: the next pass is handed the E list, nothing else.  __syncthreads() would also wait for the
                    //  acknowledgement of this pass's fire-and-forget traffic — touched / deferred list stores, open-mask atomics: one more
                    //  L2 round trip per pass on the critical chain.  A later pass that reads a word an unacknowledged atomic is about to
                    //  lower sees the older, HIGHER value: it may issue an atomicMin that loses, never skip one that would win.)
                    xr_lds_barrier();
#ifdef XR_BIG_TWO_BARRIERS          // A/B: round 4's second barrier per pass (the counters need none)
                    xr_lds_barrier();
#endif
#ifdef XR_BIG_PASS_PROBE
                    if (tid == 0) {
                        const long long pp4 = clock64();
                        long long* pc = b.phase_cycles + (int64_t)e * 8;
                        pc[5] += pp4 - pp0; pc[6] += 1; pc[7] += nE;
                        if (pp1 && pp2 && pp3) { pc[0] += pp1 - pp0; pc[1] += pp2 - pp1; pc[2] += pp3 - pp2; pc[3] += pp4 - pp3; pc[4] += 1; }
                    }
#endif
                    eb ^= 1; ec = ec1;
                }
            }
            lmin = xr_wave_min_u32(lmin);
            if ((tid & 63) == 0 && lmin != XR_DIAL_INF) atomicMin(&s_min[nx1], lmin);
            nrounds++;
#ifdef XR_PHASE_TIMING
            if (tid == XR_TIMING_TID) _ph[7] += 1;
#endif
            __syncthreads();
            cur = nx1;
        }
        XR_LAP(2);

        if (aborted) {                        // uniform: the remaining pins are charged as unreachable, nothing is traced
            if (tid == 0) { d_vio += s_remaining; status |= XR_ENV_ROUTER_ABORT | XR_ENV_UNREACHABLE; s_remaining = 0; s_target_i = -1; s_plen = 0; }
        } else
        // ---- nearest access point of an unconnected pin; ties -> lowest flat index (wave 0) --------
        if (tid < 64) {
            for (int i = tid; i < nap; i += 64) {
                if (s_ap_conn[i]) continue;
                const uint32_t w = xr_ld(&fieldg[s_ap_f[i]]);
                if (w >= XR_W_USABLE_END) continue;                    // CLEAN: never reached
                atomicMin(&s_tkey, ((unsigned long long)(w >> 2) << 32) | (unsigned)s_ap_f[i]);
            }
            __builtin_amdgcn_wave_barrier();
            const unsigned long long bestk = s_tkey;
            int best_i = -1;
            if (bestk != ~0ULL) {
                const int bf = (int)(bestk & 0xFFFFFFFFu);
                for (int i0 = 0; i0 < nap && best_i < 0; i0 += 64) {
                    const int i = i0 + tid;
                    const unsigned long long mm = __ballot(i < nap && s_ap_f[i] == bf);
                    if (mm) best_i = i0 + __ffsll((long long)mm) - 1;
                }
            }
            if (tid == 0) { s_target_i = best_i; s_plen = 0; }

            if (best_i < 0) {
                if (tid == 0) {
                    d_vio += s_remaining;
                    status |= XR_ENV_UNREACHABLE;
                    s_remaining = 0;
                }
            } else {
                // back-trace (see the LDS form: branch-free predecessor arithmetic, two hops per round trip — here the round trips
                // go to L2); a CLEAN neighbour was never reached and cannot be a predecessor
                int v = __builtin_amdgcn_readfirstlane(s_ap_f[best_i]);
                uint32_t vw = xr_ld(&fieldg[v]);
                uint32_t ux, ur, uy, uz;
                xr_divmod((uint32_t)v, uYZ, R.magic_yz, ux, ur);
                xr_divmod(ur, uZ, R.magic_z, uy, uz);
                int x = (int)ux, y = (int)uy, z = (int)uz;
                int np = 0;
                const int lvl2 = tid >= 6 && tid < 42;
                const int d1 = tid < 6 ? tid : (tid < 42 ? (tid - 6) / 6 : 6);
                const int d2 = lvl2 ? (tid - 6) % 6 : 6;
                auto pred_of = [&](int f, int cx, int cy, int cz, int d, uint32_t& len4) -> int {
                    const bool vt = (ldir >> cz) & 1u;
                    const int ddx = (d == 0) - (d == 2), ddy = (d == 3) - (d == 1), ddz = (d == 4) - (d == 5);
                    const int nx = cx + ddx, ny = cy + ddy, nz = cz + ddz;
                    const bool along = ddz != 0 || (ddx != 0 && !vt) || (ddy != 0 && vt);
                    const bool inb = d < 6 && along && (unsigned)nx < (unsigned)X && (unsigned)ny < (unsigned)Y && (unsigned)nz < (unsigned)Z;
                    const uint32_t* ctab = ddx ? s_xc : s_yc;
                    const int c0 = ddx ? cx : cy, c1 = c0 + ddx + ddy;
                    const uint32_t ca = ctab[c0 + 1], cb = ctab[(ddz ? c0 : c1) + 1];
                    len4 = ddz ? via4 : (cb > ca ? cb - ca : ca - cb);
                    return inb ? f + ddx * YZ + ddy * Z + ddz : -1;
                };
                auto claim_node = [&](int node, uint32_t w, int src, uint32_t pl4) {     // thread 0
                    if (w & 2u) { d_vio += 1; d_held += 1; }
                    pathg[np] = (uint32_t)node;
                    if (plen < b.path_cap) path[plen] = node;
                    plen++;
                    fnv_mix(h, (uint32_t)node);
                    if (src >= 4) d_via += 1; else d_wl += (int)(pl4 >> 2);
                };
                for (int nt = 0; (vw >> 2) > 0; nt++) {
                    if (nt > N) { if (tid == 0) status |= 0x100; break; }    // (hang guard: distances strictly decrease)
                    const uint32_t need4 = (vw & ~3u) - ((vw & 2u) ? pen4 : 0u) - guide_of(x, y, z);
                    uint32_t len1 = 0, len2 = 0;
                    const int u1 = pred_of(v, x, y, z, d1, len1);
                    const int x1 = x + (d1 == 0) - (d1 == 2), y1 = y + (d1 == 3) - (d1 == 1), z1 = z + (d1 == 4) - (d1 == 5);
                    const int u2 = (lvl2 && u1 >= 0) ? pred_of(u1, x1, y1, z1, d2, len2) : -1;
                    const uint32_t rw1 = xr_ld(&fieldg[u1 >= 0 ? u1 : v]), rw2 = xr_ld(&fieldg[u2 >= 0 ? u2 : v]);
                    const uint32_t uw1 = u1 >= 0 ? rw1 : XR_W_BLOCK, uw2 = u2 >= 0 ? rw2 : XR_W_BLOCK;
                    const bool use1 = (uw1 - 1u) < (XR_W_USABLE_END - 1u);
                    const bool ok1 = tid < 6 && use1 && (uw1 & ~3u) + len1 == need4;
                    const bool ok2 = use1 && (uw2 - 1u) < (XR_W_USABLE_END - 1u) &&
                                     (uw2 & ~3u) + len2 == (uw1 & ~3u) - ((uw1 & 2u) ? pen4 : 0u) - guide_of(x1, y1, z1);
                    const unsigned long long mm = __ballot(ok1), mm2 = __ballot(ok2);
                    if (mm == 0) { if (tid == 0) status |= 0x100; break; }
                    const int src = __ffsll((long long)mm) - 1;
                    const int pu = __builtin_amdgcn_readlane(u1, src);
                    const uint32_t puw = (uint32_t)__builtin_amdgcn_readlane((int)uw1, src);
                    const uint32_t pl4 = (uint32_t)__builtin_amdgcn_readlane((int)len1, src);
                    if (tid == 0) claim_node(v, vw, src, pl4);
                    np++;
                    x += (src == 0) - (src == 2);
                    y += (src == 3) - (src == 1);
                    z += (src == 4) - (src == 5);
                    v = pu; vw = puw;
                    if ((vw >> 2) == 0) break;
                    const uint32_t m6 = (uint32_t)(mm2 >> (6 + 6 * src)) & 63u;
                    if (m6 == 0) { if (tid == 0) status |= 0x100; break; }
                    const int srcb = __ffs((int)m6) - 1;
                    const int lane2 = 6 + 6 * src + srcb;
                    const int pu2 = __builtin_amdgcn_readlane(u2, lane2);
                    const uint32_t puw2 = (uint32_t)__builtin_amdgcn_readlane((int)uw2, lane2);
                    const uint32_t pl42 = (uint32_t)__builtin_amdgcn_readlane((int)len2, lane2);
                    if (tid == 0) claim_node(v, vw, srcb, pl42);
                    np++;
                    x += (srcb == 0) - (srcb == 2);
                    y += (srcb == 3) - (srcb == 1);
                    z += (srcb == 4) - (srcb == 5);
                    v = pu2; vw = puw2;
                }
                if (tid == 0 && (status & 0x100)) {
                    s_remaining = 0;
                } else if (tid == 0) {
                    if (owner[v] == 0) {
                        owner[v] = claim_val;
                        if (plen < b.path_cap) path[plen] = v;
                        plen++;
                        fnv_mix(h, (uint32_t)v);
                    }
                    s_remaining -= 1;
                    s_plen = np;
                }
            }
        }
        __syncthreads();
        XR_LAP(3);
        {
            const int ti = s_target_i;
            if (ti >= 0) {
                const short pin = s_ap_pin[ti];
                for (int i = tid; i < nap; i += nthr)
                    if (s_ap_pin[i] == pin) { s_ap_conn[i] = 1; make_source(s_ap_f[i]); }
                const int np = s_plen;
                for (int i = tid; i < np; i += nthr) {
                    const int f = (int)pathg[i];
                    make_source(f);
                    if (owner[f] == 0) owner[f] = claim_val;
                }
            }
        }
        XR_LAP(4);
    }
    // ---- leave the scratch CLEAN: reset exactly what this route (attempt) touched ----------------------------
    {
        const int nt = s_ntouched;
        for (int i = tid; i < nt; i += nthr) {
            const uint32_t f = touchg[i];
            xr_st(&fieldg[f], XR_BIG_CLEAN);
            xr_st(&openg[f >> 5], 0u);
            xr_st(&wming[f >> 5], XR_DIAL_INF);
        }
    }
    if (!XR3_ALL_ATTEMPTS || !V2 || b.maze_end_iter <= 1) break;
    // ---- XR-Maze v2: does the attempt stand?  Its path uses a node held by another net and attempts are left: rip it up ----
    if (tid == 0) s_retry = (d_held > 0 && attempt + 1 < b.maze_end_iter) ? 1 : 0;
    __syncthreads();
    const bool retry = s_retry != 0;
    for (int f = tid; f < N; f += nthr)                       // tentative claims: accepted (-a -> a) or undone (-a -> 0)
        if (owner[f] == (int16_t)-a) owner[f] = retry ? (int16_t)0 : (int16_t)a;
    if (!retry) break;
    attempt++;
    pen4 = ((uint32_t)b.pen_cost << 2) << attempt;
    if (tid == 0) { d_vio = 0; d_wl = 0; d_via = 0; plen = 0; d_held = 0; status = XR_ENV_OK; h = h0; s_ntouched = 0; s_ndefer = 0; }
    for (int i = tid; i < ng; i += nthr) s_gmin[i] = XR_DIAL_INF;
    __syncthreads();                                          // scratch CLEAN, owner grid restored: start over from the first pin
    for (int i = tid; i < nap; i += nthr) {
        const bool iso = s_ap_conn[i] == 2;
        const bool first = s_ap_pin[i] == (short)s_first_pin;
        s_ap_conn[i] = iso ? 2 : (first ? 1 : 0);
        if (first && !iso) make_source(s_ap_f[i]);
    }
    if (tid == 0) s_remaining = s_npins - 1 - n_isolated;
    // (the barriers at the top of the search loop order all of this)
    }
    if (tid == 0) {
        if (n_isolated > 0) { d_vio += n_isolated; status |= XR_ENV_UNREACHABLE; }
        xr_step_epilogue(b, e, a, d_vio, d_wl, d_via, plen, status, nrounds, h, s_ntouched);
    }
    XR_LAP(5);
    XR_TDUMP();
}
