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
// Later pins re-use the field: after a connection the path nodes and the reached pin's access points become sources
// (distance 0, open); every other value is still an upper bound.
//
// Two placements: LDS_FIELD (field + bitmasks in LDS: 38.2 KB at 24x40x9, 4 workgroups per CU) and the HBM-scratch form
// for large regions (see xr_dial_route_env_big below).
#pragma once

#define XR_DIAL_INF 0xFFFFFFFFu
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
__device__ __forceinline__ bool xr_step_prologue(const XrBatchDev& b, const int e, const int a) {
    const int tid = threadIdx.x;
    if (b.nlegal[e] == 0) {
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
    const XrRegionDev& R = b.regions[b.env_region[e]];
    bool valid = (a >= 1 && a <= R.n_nets);
    if (valid) valid = (b.legal[(int64_t)e * b.legal_words + ((a - 1) >> 6)] >> ((a - 1) & 63)) & 1ULL;
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

// Game.step bookkeeping (reference baseline/baseline_utils.py:412, :426-438) + reward, by thread 0
__device__ __forceinline__ void xr_step_epilogue(const XrBatchDev& b, const int e, const int a, int d_vio, int d_wl, int d_via,
                                                 int plen, int status, int nrounds, uint64_t h) {
    if (plen > b.path_cap) status |= XR_ENV_PATH_TRUNC;
    b.cum[3 * e + 0] += d_vio; b.cum[3 * e + 1] += d_wl; b.cum[3 * e + 2] += d_via;
    b.delta[3 * e + 0] = d_vio; b.delta[3 * e + 1] = d_wl; b.delta[3 * e + 2] = d_via;
    // reward = -1 * (violation*500 + via*4 + wirelength*0.5)   (train_DQN.py:98-99), in double
    const double s = b.w_violation * (double)d_vio + b.w_via * (double)d_via + b.w_wirelength * (double)d_wl;
    b.reward[e] = -1.0 * s;
    b.legal[(int64_t)e * b.legal_words + ((a - 1) >> 6)] &= ~(1ULL << ((a - 1) & 63));
    const int nl = b.nlegal[e] - 1;
    b.nlegal[e] = nl;
    b.done[e] = (nl == 0);
    b.status[e] = status;
    b.path_len[e] = plen;
    b.sweeps[e] = nrounds;
    fnv_mix(h, (uint32_t)a);
    fnv_mix(h, (uint32_t)d_vio); fnv_mix(h, (uint32_t)d_wl); fnv_mix(h, (uint32_t)d_via);
    fnv_mix(h, (uint32_t)plen);
    b.hash[e] = h;
    b.env_steps[e] += 1;
    atomicAdd(b.total_steps, 1ULL);
    xr_publish_record(b, e);
}

// ------------------------------------------------------------------------------------------------
// LDS form.  LDS carve (dynamic):  field u32[n_max] | open | defer | claim | wmin  (u32[mw_max] each, mw_max =
// n_max / 32 + 1) | el4x | el4y.
//   wmin[w]  lower bound of the distances of the open nodes of word w (XR_DIAL_INF: none): a round reads ONE word per
//            lane to know whether any of its ~32 nodes falls into the bucket; only those words are taken and classified.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void xr_dial_route_env(const XrBatchDev& b, const int e, const int a, char* smem) {
    __shared__ unsigned short s_ap_f[XR_MAX_AP_PER_NET];      // flat node index (the LDS form holds < 64 k nodes)
    __shared__ short s_ap_pin[XR_MAX_AP_PER_NET];
    __shared__ unsigned char s_ap_conn[XR_MAX_AP_PER_NET];
    __shared__ uint32_t s_min[3], s_bst[3];
    __shared__ unsigned long long s_tkey;
    __shared__ int s_remaining, s_target_i, s_first_pin, s_npins;

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
    const int16_t* __restrict__ node_net = b.rg_node_net + R.node_off;
    int16_t* __restrict__ owner = b.owner + (int64_t)e * b.n_max;

    uint32_t* field = reinterpret_cast<uint32_t*>(smem);
    const int mw_max = (b.n_max >> 5) + 1;
    uint32_t* s_open = field + b.n_max;
    uint32_t* s_defer = s_open + mw_max;
    uint32_t* s_claim = s_defer + mw_max;
    uint32_t* s_wmin = s_claim + mw_max;
    uint32_t* s_el4x = s_wmin + mw_max;
    uint32_t* s_el4y = s_el4x + (b.x_max + 2);

    // access points of the net and the edge tables: loads issued now, consumed after the grid build (three dependent
    // global round trips otherwise sit on the critical path of a 100 us kernel)
    const int ap_lo = b.net_csr[R.net_off + a], ap_hi = b.net_csr[R.net_off + a + 1];
    const int nap = ap_hi - ap_lo;    // 1 <= nap <= XR_MAX_AP_PER_NET (checked at load)
    int my_ap_f = 0, my_ap_pin = 0;
    if (tid < nap) { my_ap_f = b.ap_node[R.ap_off + ap_lo + tid]; my_ap_pin = b.ap_pin[R.ap_off + ap_lo + tid]; }
    uint32_t my_elx = 0, my_ely = 0;
    if (tid >= 1 && tid < X) my_elx = (uint32_t)(b.coords[R.xs_off + tid] - b.coords[R.xs_off + tid - 1]) << 2;
    if (tid >= 1 && tid < Y) my_ely = (uint32_t)(b.coords[R.ys_off + tid] - b.coords[R.ys_off + tid - 1]) << 2;

    // ---- grid build: field word of every node for THIS net.  node_net / owner rows are padded to multiples of 8
    // elements (16-byte loads); pad slots and blockages become 0.  Loads of four chunks are issued before the first use.
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
    // edge length tables (x4): el4x[i] = 4*(xs[i]-xs[i-1]), 0 at both ends
    if (tid <= X) s_el4x[tid] = my_elx;
    if (tid <= Y) s_el4y[tid] = my_ely;
    for (int i = tid + nthr; i <= X; i += nthr)
        s_el4x[i] = (i < X) ? (uint32_t)(b.coords[R.xs_off + i] - b.coords[R.xs_off + i - 1]) << 2 : 0u;
    for (int i = tid + nthr; i <= Y; i += nthr)
        s_el4y[i] = (i < Y) ? (uint32_t)(b.coords[R.ys_off + i] - b.coords[R.ys_off + i - 1]) << 2 : 0u;
    if (tid == 0) { s_first_pin = 0x7FFFFFFF; s_npins = 0; }
    __syncthreads();
    for (int i = tid; i < nap; i += nthr) {
        const int pin = i < nthr ? my_ap_pin : (int)b.ap_pin[R.ap_off + ap_lo + i];
        s_ap_f[i] = (unsigned short)(i < nthr ? my_ap_f : b.ap_node[R.ap_off + ap_lo + i]);
        s_ap_pin[i] = (short)pin;
        s_ap_conn[i] = 0;
        atomicMin(&s_first_pin, pin);
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
        if (pin == (short)s_first_pin) { s_ap_conn[i] = 1; make_source(s_ap_f[i]); }
    }
    __syncthreads();
    if (tid == 0) s_remaining = s_npins - 1;
    XR_LAP(0);

    const uint32_t via4 = (uint32_t)b.via_cost << 2;
    const uint32_t pen4 = (uint32_t)b.pen_cost << 2;
    const uint32_t delta = R.w_min * (uint32_t)b.dial_mult;
    const uint32_t uYZ = (uint32_t)YZ, uZ = (uint32_t)Z;
    int d_vio = 0, d_wl = 0, d_via = 0, plen = 0, status = XR_ENV_OK;   // thread 0 only
    int nrounds = 0;
    uint64_t h = (tid == 0) ? b.hash[e] : 0;
    int32_t* __restrict__ path = b.path + (int64_t)e * b.path_cap;

    for (;;) {
        // ---- new search: sources are open with distance 0; deferred nodes are looked at again ------------
        if (tid == 0) {
            s_min[0] = 0; s_min[1] = XR_DIAL_INF; s_min[2] = XR_DIAL_INF;
            s_bst[0] = XR_DIAL_INF; s_bst[1] = XR_DIAL_INF; s_bst[2] = XR_DIAL_INF;
            s_tkey = ~0ULL;
        }
        for (int i = tid; i < mw; i += nthr) {
            const uint32_t m = s_defer[i];
            if (m) { atomicOr(&s_open[i], m); s_defer[i] = 0; s_wmin[i] = 0u; }   // (0: a lower bound; the first scan fixes it)
        }
        __syncthreads();
        if (s_remaining <= 0) break;          // uniform: written before the barrier above
        int cur = 0;
        for (;;) {
            const int nx1 = cur == 2 ? 0 : cur + 1, nx2 = nx1 == 2 ? 0 : nx1 + 1;
            const uint32_t m = s_min[cur], best = s_bst[cur];
            if (m == XR_DIAL_INF || m > best) break;                 // uniform
            const uint32_t hi = m + delta;
            uint32_t lmin = XR_DIAL_INF;
            if (tid == 0) { s_min[nx2] = XR_DIAL_INF; s_bst[nx2] = XR_DIAL_INF; }
            // bound for the next round: smallest tentative distance of an unconnected target
            // (by the threads at the END of the workgroup: the first wave carries the words beyond one per thread)
            for (int i = nthr - 1 - tid; i < nap; i += nthr)
                if (!s_ap_conn[i]) { const uint32_t w = field[s_ap_f[i]]; if (w < XR_W_UNREACHED) atomicMin(&s_bst[nx1], w >> 2); }
            // A lane scans word wi and — where the mask has more words than the workgroup has threads — word wi + nthr in
            // the SAME pass (one 64-bit bit set), so that no wave runs the body twice per round.
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
#pragma unroll
                        for (int j = 0; j < 4; j++) w[j] = q[j] >= 0 ? field[(q[j] & 31) * mw + (q[j] < 32 ? wi : wi2)] : XR_DIAL_INF;
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            if (q[j] < 0) continue;
                            const uint32_t d = w[j] >> 2;
                            if (d >= hi) {
                                keep |= 1ULL << q[j];
                                if (q[j] < 32) kminA = d < kminA ? d : kminA; else kminB = d < kminB ? d : kminB;
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
                        len4[0] = vert ? s_el4y[y + 1] : s_el4x[x + 1];
                        nf[1] = vert ? (y > 0 ? (int)f - Z : -1) : (x > 0 ? (int)f - YZ : -1);
                        len4[1] = vert ? s_el4y[y] : s_el4x[x];
                        nf[2] = ((int)z + 1 < Z) ? (int)f + 1 : -1; len4[2] = via4;
                        nf[3] = (z > 0) ? (int)f - 1 : -1;          len4[3] = via4;
#pragma unroll
                        for (int k = 0; k < 4; k++) wn[k] = nf[k] >= 0 ? field[nf[k]] : XR_W_BLOCK;
                        bool refused = false;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const uint32_t cand4 = d4 + len4[k] + ((wn[k] & 2u) ? pen4 : 0u);
                            cw[k] = cand4 | (wn[k] & 3u);
                            // blockage, a distance that does not exist (>= XR_DIST_CAP, spec), or no improvement: nothing to do
                            bool go = wn[k] != XR_W_BLOCK && cand4 < XR_W_USABLE_END && cw[k] < wn[k];
                            if (go && (cand4 >> 2) > best) { refused = true; go = false; }      // bound pruning
                            if (!go) cw[k] = XR_DIAL_INF;
                        }
                        // the four atomics are issued back to back; their results are looked at afterwards
#pragma unroll
                        for (int k = 0; k < 4; k++) old[k] = cw[k] != XR_DIAL_INF ? atomicMin(&field[nf[k]], cw[k]) : 0u;
                        int next_f = -1;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            if (cw[k] < old[k]) {                      // lowered
                                if (XR_DIAL_CHAIN && next_f < 0 && (cw[k] >> 2) < hi) { next_f = nf[k]; continue; }
                                uint32_t oq, orr;                      // the neighbour becomes open
                                xr_divmod((uint32_t)nf[k], umw, magic_mw, oq, orr);
                                atomicOr(&s_open[orr], 1u << oq);
                                atomicMin(&s_wmin[orr], cw[k] >> 2);
                                lmin = (cw[k] >> 2) < lmin ? (cw[k] >> 2) : lmin;
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
            XR_LAP(1);
            lmin = xr_wave_min_u32(lmin);
            if ((tid & 63) == 0 && lmin != XR_DIAL_INF) atomicMin(&s_min[nx1], lmin);
            nrounds++;
#ifdef XR_PHASE_TIMING
            if (tid == XR_TIMING_TID) _ph[7] += 1;
#endif
            __syncthreads();
            XR_LAP(6);
            cur = nx1;
        }
        XR_LAP(6);

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
                // direction order, build_3Dgrid.py:127).  Lanes 0..5 test one direction each; ballot + ffs picks the
                // first match (a wave-uniform lane id, so the winner's values are read with v_readlane, no LDS trip).
                // The field is only READ here; claimed nodes are zeroed afterwards.
                int v = __builtin_amdgcn_readfirstlane((int)s_ap_f[best_i]);
                uint32_t vw = field[v];
                uint32_t ux, ur, uy, uz;
                xr_divmod((uint32_t)v, uYZ, R.magic_yz, ux, ur);
                xr_divmod(ur, uZ, R.magic_z, uy, uz);
                int x = (int)ux, y = (int)uy, z = (int)uz;
                while ((vw >> 2) > 0) {
                    const uint32_t need4 = (vw & ~3u) - ((vw & 2u) ? pen4 : 0u);   // pred distance + edge, x4
                    const bool vert = (ldir >> z) & 1u;
                    int u = -1;
                    uint32_t len4 = 0;
                    switch (tid) {
                    case 0: if (!vert && x + 1 < X) { u = v + YZ; len4 = s_el4x[x + 1]; } break;   // E
                    case 1: if (vert && y > 0)      { u = v - Z;  len4 = s_el4y[y]; } break;       // S
                    case 2: if (!vert && x > 0)     { u = v - YZ; len4 = s_el4x[x]; } break;       // W
                    case 3: if (vert && y + 1 < Y)  { u = v + Z;  len4 = s_el4y[y + 1]; } break;   // N
                    case 4: if (z + 1 < Z)          { u = v + 1;  len4 = via4; } break;           // U
                    case 5: if (z > 0)              { u = v - 1;  len4 = via4; } break;           // D
                    default: break;
                    }
                    uint32_t uw = XR_W_BLOCK;
                    bool ok = false;
                    if (u >= 0) {
                        uw = field[u];
                        ok = (uw - 1u) < (XR_W_USABLE_END - 1u) && (uw & ~3u) + len4 == need4;
                    }
                    const unsigned long long mm = __ballot(ok);
                    if (mm == 0) { if (tid == 0) status |= 0x100; break; }     // inconsistent field: cannot happen
                    const int src = __ffsll((long long)mm) - 1;                // wave-uniform
                    const int pu = __builtin_amdgcn_readlane(u, src);
                    const uint32_t puw = (uint32_t)__builtin_amdgcn_readlane((int)uw, src);
                    const uint32_t pl4 = (uint32_t)__builtin_amdgcn_readlane((int)len4, src);
                    if (tid == 0) {                 // claim v
                        uint32_t cq, cr;
                        xr_divmod((uint32_t)v, umw, magic_mw, cq, cr);
                        if (vw & 2u) d_vio += 1;
                        atomicOr(&s_claim[cr], 1u << cq);
                        if (plen < b.path_cap) path[plen] = v;
                        plen++;
                        fnv_mix(h, (uint32_t)v);
                        if (src >= 4) d_via += 1; else d_wl += (int)(pl4 >> 2);
                    }
                    x += (src == 0) - (src == 2);
                    y += (src == 3) - (src == 1);
                    z += (src == 4) - (src == 5);
                    v = pu; vw = puw;
                }
                if (tid == 0 && (status & 0x100)) {
                    s_remaining = 0;              // never taken on a consistent field; avoids spinning
                } else if (tid == 0) {
                    // terminal node of the component: claimed (and recorded) only if nobody holds it yet
                    if (owner[v] == 0) {
                        owner[v] = (int16_t)a;
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
                            if (owner[f] == 0) owner[f] = (int16_t)a;
                        }
                    }
                }
            }
        }
        XR_LAP(4);
        // (the barrier at the top of the loop orders these writes before the next search / the exit test)
    }

    if (tid == 0) xr_step_epilogue(b, e, a, d_vio, d_wl, d_via, plen, status, nrounds, h);
    XR_LAP(5);
    XR_TDUMP();
}
