// xr_kernels.hip — gfx950 (MI355X, wave64) kernels of the env hot path.
//
//   xr_ingest_kernel      packed node records -> compact state (node_net, owner0)      HBM sweep
//   xr_reset_kernel       Game.reset bookkeeping for masked envs                        HBM copy
//   xr_route_kernel       Game.step: grid build -> XR-Maze v1 maze route -> claim path -> metrics
//                         one workgroup per env; distance field resident in LDS (<= ~32k nodes)
//                         or in a per-env HBM scratch (larger regions)
//                         + (xr_batch_step_observe) the observation of the new state by the same workgroup
//   xr_order_kernel       whole-order re-route (A3C / MCTS contracts): reset + route a list of nets, one launch
//   xr_obs_kernel         build_3Dgrid: compact state -> fp32 [2+7K, Z, Y, X] observation, streaming
//   xr_plan_kernel + xr_step_queue_kernel   the step with its observation, default form (XR_OBS_QUEUE): one persistent
//                         launch draining route tasks and net-plane units
//   xr_netplane_kernel, xr_netplane_stream_kernel   the writer of the split form (XR_OBS_SPLIT)
//   xr_random_action_kernel
//
// Integer / index work throughout: no MFMA.  What matters here is coalescing (every sweep is
// unit-stride in the reference's own flat order f=(x*Y+y)*Z+z), LDS residency of the distance
// field, and wave64-wide relaxation.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "xr_device.h"
#include "../../include/xroute_hip.h"

#define XR_FNV_PRIME 0x100000001b3ULL

__device__ __forceinline__ void fnv_mix(uint64_t& h, uint32_t w) { h = (h ^ (uint64_t)w) * XR_FNV_PRIME; }

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// ------------------------------------------------------------------------------------------------
// ingest: one thread per node of every region.  Net/used semantics follow handle_messange
// (reference baseline/baseline_utils.py:23-39): Net = net+1 for ACCESS, -1 BLOCKAGE, 0 NORMAL.
// ------------------------------------------------------------------------------------------------
__global__ void xr_ingest_kernel(const uint32_t* __restrict__ rec, int16_t* __restrict__ node_net,
                                 int16_t* __restrict__ owner0, int64_t total) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const uint32_t r = rec[i];
        const uint32_t t = XR_REC_TYPE(r);
        const int net1 = (int)XR_REC_NET1(r);
        int16_t nn = 0;
        if (t == XR_TYPE_ACCESS) nn = (int16_t)net1;
        else if (t == XR_TYPE_BLOCKAGE) nn = -1;
        int16_t ow = 0;
        if (XR_REC_USED(r)) ow = (t == XR_TYPE_ACCESS) ? (int16_t)net1 : (int16_t)XR_OWNER_FOREIGN;
        node_net[i] = nn;
        owner0[i] = ow;
    }
}

// ------------------------------------------------------------------------------------------------
// reset of one env by its workgroup (Game.reset bookkeeping, reference
// baseline/baseline_utils.py:466-473; region rotation examples/launch_training.py:33-54)
// ------------------------------------------------------------------------------------------------
// packed result record of env e (thread 0, after it has written the per-field arrays)
__device__ __forceinline__ void xr_publish_record(const XrBatchDev& b, int e) {
    XrStepRecord* r = b.records + e;
    r->reward = b.reward[e];
    r->delta[0] = b.delta[3 * e]; r->delta[1] = b.delta[3 * e + 1]; r->delta[2] = b.delta[3 * e + 2];
    r->cum[0] = b.cum[3 * e]; r->cum[1] = b.cum[3 * e + 1]; r->cum[2] = b.cum[3 * e + 2];
    r->nlegal = b.nlegal[e];
    r->env_steps = (int32_t)b.env_steps[e];
    r->path_len = b.path_len[e];
    r->done = b.done[e]; r->pad = 0;
    r->status = (uint16_t)b.status[e];
}

__device__ void xr_env_reset(const XrBatchDev& b, int e, int rotate, int extra_status) {
    __shared__ int s_region;
    const int tid = threadIdx.x;
    if (tid == 0) {
        int r = b.env_region[e];
        if (rotate) {
            int rc = b.env_replay[e];
            if (rc == b.max_route_count) {          // launch_training.py:37-46
                rc = 1;
                r = (int)(((int64_t)r + b.n_envs) % b.n_regions);
            } else {
                rc += 1;
            }
            b.env_replay[e] = rc;
            b.env_region[e] = r;
        }
        s_region = r;
    }
    __syncthreads();
    const XrRegionDev R = b.regions[s_region];
    const int16_t* __restrict__ src = b.rg_owner0 + R.node_off;
    int16_t* __restrict__ dst = b.owner + (int64_t)e * b.n_max;
    // node_off and n_max are multiples of 8 elements: 16-byte vector copy
    const int nvec = R.N >> 3;
    const int4* s4 = reinterpret_cast<const int4*>(src);
    int4* d4 = reinterpret_cast<int4*>(dst);
    for (int i = tid; i < nvec; i += blockDim.x) d4[i] = s4[i];
    for (int i = (nvec << 3) + tid; i < R.N; i += blockDim.x) dst[i] = src[i];
    for (int w = tid; w < b.legal_words; w += blockDim.x)
        b.legal[(int64_t)e * b.legal_words + w] = b.legal0[R.legal0_off + w];
    if (tid == 0) {
        b.nlegal[e] = R.nlegal0;
        b.cum[3 * e + 0] = R.m0[0]; b.cum[3 * e + 1] = R.m0[1]; b.cum[3 * e + 2] = R.m0[2];
        b.delta[3 * e + 0] = 0; b.delta[3 * e + 1] = 0; b.delta[3 * e + 2] = 0;
        b.reward[e] = -0.0;
        b.done[e] = (R.nlegal0 == 0);
        b.status[e] = extra_status;
        b.path_len[e] = 0;
        b.sweeps[e] = 0;
        b.touched[e] = 0;
        xr_publish_record(b, e);
    }
}

// ------------------------------------------------------------------------------------------------
// ingest of a NEW state of every env that an external simulator produced (BASELINE config 2: "grid-build + reward only"): the client
// half of Game.step (reference baseline/baseline_utils.py:420-438) without the route — what a `Request` carries per env (occupancy of
// every node, the nets still to route, the cumulative metrics) becomes the env's state; the metric deltas are new - previous cumulative
// values (:426-428), the reward is the trainers' expression in double (baseline/DQN/train_DQN.py:98-99), done = no nets left (:435-436).
// One workgroup per env: a 16-byte vector copy of the owner row (2·N read + 2·N written), the legal bitmask cut to the region's nets.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) xr_ingest_state_kernel(XrBatchDev b, const int16_t* __restrict__ owner_in, const uint64_t* __restrict__ legal_in,
                                                              const int32_t* __restrict__ cum_in) {
    __shared__ int s_cnt[4];
    const int e = blockIdx.x, tid = threadIdx.x;
    const XrRegionDev R = b.regions[b.env_region[e]];
    const int4* s4 = reinterpret_cast<const int4*>(owner_in + (int64_t)e * b.n_max);
    int4* d4 = reinterpret_cast<int4*>(b.owner + (int64_t)e * b.n_max);
    const int nvec = (R.N + 7) >> 3;                         // rows are padded to multiples of 8 elements (n_max)
    for (int i = tid; i < nvec; i += blockDim.x) d4[i] = s4[i];
    int cnt = 0;
    for (int w = tid; w < b.legal_words; w += blockDim.x) {
        const int lo = w * 64, K = R.n_nets;
        const uint64_t allowed = K >= lo + 64 ? ~0ULL : (K > lo ? ((1ULL << (K - lo)) - 1ULL) : 0ULL);
        const uint64_t v = legal_in[(int64_t)e * b.legal_words + w] & allowed;      // (a bit beyond the region's nets names no net: dropped)
        b.legal[(int64_t)e * b.legal_words + w] = v;
        cnt += __popcll(v);
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
    if ((tid & 63) == 0) s_cnt[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0) {
        const int nl = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        const int c0 = cum_in[3 * e], c1 = cum_in[3 * e + 1], c2 = cum_in[3 * e + 2];
        const int d_vio = c0 - b.cum[3 * e], d_wl = c1 - b.cum[3 * e + 1], d_via = c2 - b.cum[3 * e + 2];
        b.cum[3 * e] = c0; b.cum[3 * e + 1] = c1; b.cum[3 * e + 2] = c2;
        b.delta[3 * e] = d_vio; b.delta[3 * e + 1] = d_wl; b.delta[3 * e + 2] = d_via;
        const double s = b.w_violation * (double)d_vio + b.w_via * (double)d_via + b.w_wirelength * (double)d_wl;
        b.reward[e] = -1.0 * s;
        b.nlegal[e] = nl;
        b.done[e] = (nl == 0);
        b.status[e] = XR_ENV_OK;
        b.path_len[e] = 0;
        b.env_steps[e] += 1;
        xr_publish_record(b, e);
    }
}

__global__ void xr_reset_kernel(XrBatchDev b, const uint8_t* __restrict__ mask, int rotate) {
    const int e = blockIdx.x;
    if (mask && !mask[e]) return;
    xr_env_reset(b, e, rotate, XR_ENV_OK);
}

// ------------------------------------------------------------------------------------------------
// route: one workgroup per env.
//
// XR-Maze v1.  The distance field of a multi-source shortest-path problem is unique,
// so any relaxation order that reaches the fixpoint gives the oracle's (Dijkstra's) field; the
// target choice and the back-trace are deterministic functions of that field.
//
// Relaxation = line-segment sweeps ("fast sweeping" on the routing graph).  A *line* is a track (one
// preferred-direction row of one layer: x-lines on horizontal layers, y-lines on vertical layers) or
// a column (the via chain of one (x,y)); a worklist *item* is a (line, chunk of XR_CH nodes) pair.  One thread
// takes one item and runs a forward and a backward Gauss-Seidel pass that start at that chunk, staged through
// registers 8 nodes at a time, and run on along the line for as long as nodes keep getting lowered (xr_seg_pass),
// so a distance still travels any straight run in ONE visit.  Edge lengths come from two small LDS tables
// (4*(xs[i]-xs[i-1]), same for ys): a wave's lanes read one address.
//
// Which items to visit: a worklist.  A node lowered by a pass marks the item of the one other line through that
// node dirty (atomic OR into an LDS bitmask); every iteration compacts the dirty bitmask into dense per-kind
// worklists (kinds start on wave boundaries, so waves are full and uniform) and processes them.  A search starts
// with only the items through its source nodes dirty, so work follows the wavefront instead of sweeping the
// whole region, and a visit costs what the affected segment costs, not what the line costs.
//
// Pruning: nothing above `bound` (= best distance of any unconnected target so far) is written; every
// node with true distance <= the final target distance still gets its exact value (induction along
// its shortest path), so the target choice and the back-trace are unchanged.  An item in which a
// candidate was refused by the bound is remembered (`deferred`) and re-dirtied when the next search of
// the same net resets the bound.  Later pins re-use the field: every value is still an upper bound
// once the new path nodes are set to 0, so relaxation continues instead of restarting.
//
// Field word (one u32 per node, nothing else per node):
//     w = (distance << 2) | (held << 1) | 1          for every real node; a blockage is w = 0
//     held = node is held by another net (drc penalty + violation)
//     0xFFFFFFFD | (held << 1) = unreached.  Distances >= XR_DIST_CAP (0x07F00000) do not exist (spec,
//     mirrored by the oracle): candidates are capped there, which makes u32 wrap-around impossible.
//
// Two placements of the same code (template LDS_DIST): field + worklists (aliased by the claim bitmask) in LDS
// (regions up to ~38 k field words: 39.9 KB per workgroup at 24x40x9, 4 workgroups per CU), or in per-env HBM scratch
// with only the item bitmasks and edge tables in LDS (larger regions: stores only lowered nodes, loads chunks on demand).
// ------------------------------------------------------------------------------------------------
#ifdef XR_PHASE_TIMING
#ifndef XR_TIMING_TID
#define XR_TIMING_TID 0           // the thread whose cycle counts are recorded
#endif
#define XR_T0() long long _t = (threadIdx.x == XR_TIMING_TID) ? clock64() : 0; long long _ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define XR_LAP(k) do { if (threadIdx.x == XR_TIMING_TID) { const long long _n = clock64(); _ph[k] += _n - _t; _t = _n; } } while (0)
#define XR_TDUMP() do { if (threadIdx.x == XR_TIMING_TID) for (int _k = 0; _k < 8; _k++) b.phase_cycles[(int64_t)e * 8 + _k] += _ph[_k]; } while (0)
#else
#define XR_T0() do {} while (0)
#define XR_LAP(k) do {} while (0)
#define XR_TDUMP() do {} while (0)
#endif
#define XR_CH 8                       // generic line chunk held in registers
#define XR_W_USABLE_END 0x1FC00000u   // (XR_DIST_CAP << 2, XR_DIST_CAP = 0x07F00000): predecessors must be below this
#define XR_W_UNREACHED 0xFFFFFFFDu    // | held << 1
#define XR_W_BLOCK 0u                 // blockage (and padded register slots)

// One Gauss-Seidel pass along (a segment of) a line of L nodes: node i lives at field[ix(i)].
// PLANAR: el4[i] = 4 * distance between node i-1 and node i of the line (LDS table);
// else constant edge length len4c (via chain).
// CH nodes are staged in registers per chunk.  The loop-carried dependency per node is three VALU ops
// (saturating add -> compare -> select); everything else depends only on the loaded words:
//     add_j  = 4*len + (held ? 4*pen : 0) + flags_j - flags_{j-1}      (blocked node: 0xFFFFFFFF)
//     cand   = sat_add(w_{j-1}, add_j)                                  (= 4*d_cand + flags_j)
//     lim_j  = min(w_j, (bound+1) << 2)                                 (blocked node in registers: 0xFFFFFFFF)
//     w_j    = cand < lim_j ? cand : w_j
// An unreached or blocked predecessor saturates the add, which can never pass `lim`.  The store is an atomic min
// (concurrent passes over crossing lines, or over other chunks of the same line, lose no update).
// EXACT: L == CH, single chunk, no bounds handling.
//
// Worklist items are (line, chunk) pairs, chunk c = nodes [c*CH, (c+1)*CH) of the line.
// The pass starts at chunk c0 with the node just before it (in the direction of travel) as predecessor and enters the
// next chunk only if that chunk's first node is lowered by the boundary node of the chunk just finished: a node that
// is not lowered passes nothing new on (whatever lowered it earlier queued its own item).  The fixpoint is the one
// the full-line pass reaches; only the visiting order differs.
// `mark(start, bits)`: lowered nodes of one chunk, bit k <-> node start + k (natural order in both directions).
// `defer(c)`: a candidate was refused in chunk c only because of the bound.
// COND_STORE (field in HBM scratch): only lowered nodes are stored — an L2 atomic per node is what bounds the
// large-region variant; in LDS the unconditional min is cheaper than the branch.
template <bool FWD, bool PLANAR, int CH, bool EXACT, bool COND_STORE, class IndexFn, class MarkFn, class DeferFn>
__device__ __forceinline__ void xr_seg_pass(uint32_t* __restrict__ field, const uint32_t* __restrict__ el4, IndexFn ix,
                                            int L, uint32_t len4c, uint32_t pen4, uint32_t boundw1, int c0,
                                            MarkFn mark, DeferFn defer) {
    uint32_t prev = 0xFFFFFFFFu;       // "no predecessor": saturates
    uint32_t pfl = 3u;
    if (!EXACT) {
        const int ip = FWD ? c0 * CH - 1 : (c0 + 1) * CH;
        if (FWD ? (ip >= 0) : (ip < L)) {
            const uint32_t pw = field[ix(ip)];
            if (pw != XR_W_BLOCK) { prev = pw; pfl = pw & 3u; }
        }
    }
    uint32_t w[CH], el[CH];
    // slot j of a chunk is node c*CH + j in both directions; the chain walks j upwards (FWD) or downwards
    auto load = [&](int c, uint32_t (&ww)[CH], uint32_t (&ee)[CH]) {
        const int i0 = c * CH;
        const bool full = EXACT || (i0 + CH <= L);
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const int i = i0 + j;
            const bool in = full || (i < L);
            ww[j] = in ? field[ix(i)] : XR_W_BLOCK;
            ee[j] = PLANAR ? (in ? el4[FWD ? i : i + 1] : 0u) : len4c;
        }
    };
    const int nch = EXACT ? 1 : (L + CH - 1) / CH;
    int c = EXACT ? 0 : c0;
    load(c, w, el);
    for (;;) {
        uint32_t wnx[CH], enx[CH];
        const int cn = FWD ? c + 1 : c - 1;
        const bool more = !EXACT && (FWD ? (cn < nch) : (cn >= 0));
        // LDS: the next chunk's loads are issued before this chunk's arithmetic (software pipeline).  HBM scratch:
        // the kernel is bound by L2 transactions, not latency, so the next chunk is only loaded when it is visited
        if (more && !COND_STORE) load(cn, wnx, enx);
        const int i0 = c * CH;
        const bool full = EXACT || (i0 + CH <= L);
        uint32_t cm = 0, refused = 0;
#pragma unroll
        for (int jj = 0; jj < CH; jj++) {
            const int j = FWD ? jj : CH - 1 - jj;
            const int i = i0 + j;
            const bool in = full || (i < L);
            const uint32_t cw = w[j];
            const bool blk = (cw == XR_W_BLOCK);
            const uint32_t cwr = blk ? 0xFFFFFFFFu : cw;
            const uint32_t fl = cwr & 3u;
            const uint32_t add = blk ? 0xFFFFFFFFu : (__umul24((cw >> 1) & 1u, pen4) + el[j] + fl - pfl);
            const uint32_t lim = min(cwr, boundw1);
            const uint32_t cand = __builtin_elementwise_add_sat(prev, add);
            const bool acc = cand < lim;
            const uint32_t wn = acc ? cand : cwr;
            refused |= (cand < cwr && !acc && cand < 0xF0000000u) ? 1u : 0u;
            cm |= acc ? (1u << j) : 0u;
            if (COND_STORE ? acc : in) atomicMin(&field[ix(i)], wn);      // (acc implies a real node)
            prev = wn;
            pfl = fl;
        }
        if (cm) mark(i0, cm);
        if (refused) defer(c);
        if (!more) break;
        // continue into the next chunk only if its first node (in the direction of travel) is lowered by the boundary
        // node of this one; if it is not, nothing beyond it can change either
        {
            const int jn = FWD ? 0 : CH - 1;
            const int in = cn * CH + jn;                   // always a real node
            const uint32_t nw = COND_STORE ? field[ix(in)] : wnx[jn];
            if (nw == XR_W_BLOCK) break;
            const uint32_t nel = PLANAR ? (COND_STORE ? el4[FWD ? in : in + 1] : enx[jn]) : len4c;
            const uint32_t nadd = __umul24((nw >> 1) & 1u, pen4) + nel + (nw & 3u) - pfl;
            const uint32_t ncand = __builtin_elementwise_add_sat(prev, nadd);
            if (!(ncand < min(nw, boundw1))) {
                if (ncand < nw && ncand < 0xF0000000u) defer(cn);      // refused only because of the bound
                break;
            }
        }
        c = cn;
        if (COND_STORE) load(c, w, el);
        else {
#pragma unroll
            for (int j = 0; j < CH; j++) { w[j] = wnx[j]; el[j] = enx[j]; }
        }
    }
}

// OR a run of up to 32 consecutive bits (ids id0 + k for every set bit k of `bits`) into a bitmask
__device__ __forceinline__ void xr_or_run(uint32_t* mask, int id0, uint32_t bits) {
    const int wdx = id0 >> 5, sh = id0 & 31;
    const uint32_t m0 = bits << sh;
    const uint32_t m1 = sh ? (bits >> (32 - sh)) : 0u;
    if (m0) atomicOr(&mask[wdx], m0);
    if (m1) atomicOr(&mask[wdx + 1], m1);
}

// Field layouts.
//  PACKED_XYZ (used by both kernel variants): l = x*SX + y*SY + z with SY = Z|1, SX = (Y*SY)|1 (odd strides):
//    lanes of a wave hold consecutive lines, so wave accesses have odd word strides: LDS bank-conflict free.
//  LAYER_MAJOR (HBM-scratch variant only, compile with -DXR_SCRATCH_LAYER_MAJOR=1):
//    l = z*X*Y + (layer z vertical ? x*Y + y : y*X + x): every track unit-stride, via columns gather one word per
//    layer plane.  Measured SLOWER on BASELINE config 5 (256x256x12: 85 vs 67 ms per 64-env launch): the via
//    columns are more than half of the line visits and become 12 cache lines each.  Kept for A/B runs.
#ifndef XR_SCRATCH_LAYER_MAJOR
#define XR_SCRATCH_LAYER_MAJOR 0
#endif
template <bool LDS_DIST_>
struct XrLayout {
    static constexpr bool LDS_DIST = LDS_DIST_ || !XR_SCRATCH_LAYER_MAJOR;   // true: packed-xyz index math
    int X, Y, Z, SX, SY, XY;
    uint32_t ldir;
    __device__ __forceinline__ XrLayout(int x, int y, int z, uint32_t ld) : X(x), Y(y), Z(z), ldir(ld) {
        SY = Z | 1; SX = (Y * SY) | 1; XY = X * Y;
    }
    __device__ __forceinline__ int size() const { return LDS_DIST ? X * SX : Z * XY; }
    __device__ __forceinline__ int idx(int x, int y, int z) const {
        if (LDS_DIST) return x * SX + y * SY + z;
        return z * XY + (((ldir >> z) & 1u) ? x * Y + y : y * X + x);
    }
    __device__ __forceinline__ void decode(int l, int& x, int& y, int& z) const {
        if (LDS_DIST) { x = l / SX; const int r = l - x * SX; y = r / SY; z = r - y * SY; }
        else {
            z = l / XY; const int r = l - z * XY;
            if ((ldir >> z) & 1u) { x = r / Y; y = r - x * Y; } else { y = r / X; x = r - y * X; }
        }
    }
};

// ZCH: 0 = generic column pass (chunks of XR_CH); 9 / 12 = every region of the batch has exactly that
// many layers (single exact chunk).
template <bool LDS_DIST, int ZCH>
__device__ __forceinline__ void xr_route_env(const XrBatchDev& b, const int e, const int a, char* smem) {
    // field index of each access point: 16 bits are enough for any field that fits LDS (<= 40 k words)
    using ApIndex = typename std::conditional<LDS_DIST, unsigned short, int>::type;
    __shared__ ApIndex s_ap_l[XR_MAX_AP_PER_NET];
    __shared__ short s_ap_pin[XR_MAX_AP_PER_NET];
    __shared__ unsigned char s_ap_conn[XR_MAX_AP_PER_NET];
    __shared__ unsigned char s_hl[XR_MAX_LAYERS], s_vl[XR_MAX_LAYERS];
    __shared__ unsigned short s_ztrk[XR_MAX_LAYERS];  // first track id of layer z
    __shared__ int s_remaining, s_target_i;
    __shared__ uint32_t s_bound;
    __shared__ int s_cnt[2][3];                        // worklist sizes (H, V, C), double-buffered by parity

    const int tid = threadIdx.x;
    const int nthr = blockDim.x;

    // vector-env autoreset: an env that was done is re-initialised by this step (uniform branch)
    if (b.nlegal[e] == 0) {
        if (b.auto_reset) {
            xr_env_reset(b, e, 1, XR_ENV_WAS_RESET);
        } else if (tid == 0) {
            b.status[e] = XR_ENV_BAD_ACTION;
            b.delta[3 * e] = 0; b.delta[3 * e + 1] = 0; b.delta[3 * e + 2] = 0;
            b.reward[e] = -0.0; b.path_len[e] = 0; b.sweeps[e] = 0;
            xr_publish_record(b, e);
        }
        return;
    }

    const XrRegionDev R = b.regions[b.env_region[e]];
    bool valid = (a >= 1 && a <= R.n_nets);
    if (valid) valid = (b.legal[(int64_t)e * b.legal_words + ((a - 1) >> 6)] >> ((a - 1) & 63)) & 1ULL;
    if (!valid) {   // the reference never checks this client-side; here: flagged no-op
        if (tid == 0) {
            b.status[e] = XR_ENV_BAD_ACTION;
            b.delta[3 * e] = 0; b.delta[3 * e + 1] = 0; b.delta[3 * e + 2] = 0;
            b.reward[e] = -0.0; b.path_len[e] = 0; b.sweeps[e] = 0;
            xr_publish_record(b, e);
        }
        return;
    }

    XR_T0();
    const int X = R.X, Y = R.Y, Z = R.Z, N = R.N;
    const XrLayout<LDS_DIST> lay(X, Y, Z, R.ldir_mask);
    const int NL = lay.size();                    // field size in words (padded in the LDS layout)
    const int ncol = X * Y;
    const uint32_t ldir = R.ldir_mask;
    const int nv_layers = __popc(ldir & (Z >= 32 ? 0xFFFFFFFFu : ((1u << Z) - 1u)));
    const int nh_layers = Z - nv_layers;
    const int tracks_h = nh_layers * Y;            // lines along x, one per (y, horizontal layer)
    const int tracks_v = nv_layers * X;            // lines along y, one per (x, vertical layer)
    const int ntracks = tracks_h + tracks_v;
    // worklist items are (line, chunk of XR_CH nodes) pairs.  Item bitmask layout:
    //   [0, itH) x-tracks * chH | [itV0, itC0) y-tracks * chV | [itC0, nbits) columns (x*Y + y) * chC
    const int chH = (X + XR_CH - 1) / XR_CH, chV = (Y + XR_CH - 1) / XR_CH;
    const int chC = (ZCH > 0) ? 1 : (Z + XR_CH - 1) / XR_CH;
    const int itH = tracks_h * chH, itV = tracks_v * chV, itC = ncol * chC;
    const int itV0 = itH, itC0 = itH + itV;
    const int nbits = itC0 + itC;
    const int nlw = (nbits + 31) >> 5;
    (void)ntracks;
    const int16_t* __restrict__ node_net = b.rg_node_net + R.node_off;
    int16_t* __restrict__ owner = b.owner + (int64_t)e * b.n_max;

    // carve (LDS variant):   field u32[n_lds] | el4x | el4y | dirty0 | dirty1 | deferred u32[lw_max] each |
    //                        lists u16[lines_max], aliased by the claim bitmask u32[n_lds/32+1] (the lists are dead
    //                        while a path is traced, the claim bitmask is dead while lines are relaxed)
    // (large-region variant): field, claim bitmask and the worklists live in per-env HBM scratch; the edge tables
    //                        and the three item bitmasks stay in LDS
    uint32_t* field;
    uint32_t* s_claim;
    uint32_t* s_el4x;
    const int claim_words = (NL + 31) >> 5;
    if (LDS_DIST) {
        field = reinterpret_cast<uint32_t*>(smem);
        s_claim = nullptr;                       // set below (aliases the lists)
        s_el4x = field + b.n_lds;
    } else {
        field = b.dist_scratch + (int64_t)e * b.n_lds;
        s_claim = reinterpret_cast<uint32_t*>(b.cls_scratch + (int64_t)e * b.n_lds);   // n_lds bytes >= bitmask
        s_el4x = reinterpret_cast<uint32_t*>(smem);
    }
    uint32_t* s_el4y = s_el4x + (b.x_max + 2);
    uint32_t* s_dirty0 = s_el4y + (b.y_max + 2);
    uint32_t* s_dirty1 = s_dirty0 + b.lw_max;
    uint32_t* s_defer = s_dirty1 + b.lw_max;
    unsigned short* s_list = LDS_DIST ? reinterpret_cast<unsigned short*>(s_defer + b.lw_max)
                                      : b.list_scratch + (int64_t)e * b.lines_max;
    if (LDS_DIST) s_claim = reinterpret_cast<uint32_t*>(s_list);
    unsigned short* s_listH = s_list;
    unsigned short* s_listV = s_list + itV0;
    unsigned short* s_listC = s_list + itC0;

    // ---- grid build: field word of every node for THIS net (coalesced 16-byte loads of the state) --
    {
        const int nchunk = (N + 7) >> 3;
        for (int ci = tid; ci < nchunk; ci += nthr) {
            const int f0 = ci << 3;
            int pn[4], po[4];          // 8 x int16 each, packed
            if (f0 + 8 <= N) {
                const int4 vn = *reinterpret_cast<const int4*>(node_net + f0);
                const int4 vo = *reinterpret_cast<const int4*>(owner + f0);
                pn[0] = vn.x; pn[1] = vn.y; pn[2] = vn.z; pn[3] = vn.w;
                po[0] = vo.x; po[1] = vo.y; po[2] = vo.z; po[3] = vo.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int f = f0 + 2 * j;
                    const int n0 = (f < N) ? (int)(unsigned short)node_net[f] : 0xFFFF;
                    const int n1 = (f + 1 < N) ? (int)(unsigned short)node_net[f + 1] : 0xFFFF;
                    const int o0 = (f < N) ? (int)(unsigned short)owner[f] : 0;
                    const int o1 = (f + 1 < N) ? (int)(unsigned short)owner[f + 1] : 0;
                    pn[j] = n0 | (n1 << 16);
                    po[j] = o0 | (o1 << 16);
                }
            }
            int col = f0 / Z, z = f0 - col * Z;
            int x = col / Y, y = col - x * Y;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (f0 + j < N) {
                    const int nn = (int)(short)((j & 1) ? (pn[j >> 1] >> 16) : (pn[j >> 1] & 0xFFFF));
                    const int ow = (int)(short)((j & 1) ? (po[j >> 1] >> 16) : (po[j >> 1] & 0xFFFF));
                    uint32_t w;
                    if (nn == -1) w = XR_W_BLOCK;
                    else w = XR_W_UNREACHED | (((ow != 0 && ow != a) || (nn > 0 && nn != a)) ? 2u : 0u);   // bit 0: real node
                    field[lay.idx(x, y, z)] = w;
                }
                if (++z == Z) { z = 0; if (++y == Y) { y = 0; ++x; } }
            }
        }
    }
    if (!LDS_DIST) for (int i = tid; i < claim_words; i += nthr) s_claim[i] = 0;
    for (int i = tid; i < nlw; i += nthr) { s_dirty0[i] = 0; s_dirty1[i] = 0; s_defer[i] = 0; }
    // edge length tables (x4): el4x[i] = 4*(xs[i]-xs[i-1]), 0 at both ends
    for (int i = tid; i <= X; i += nthr)
        s_el4x[i] = (i >= 1 && i < X) ? (uint32_t)(b.coords[R.xs_off + i] - b.coords[R.xs_off + i - 1]) << 2 : 0u;
    for (int i = tid; i <= Y; i += nthr)
        s_el4y[i] = (i >= 1 && i < Y) ? (uint32_t)(b.coords[R.ys_off + i] - b.coords[R.ys_off + i - 1]) << 2 : 0u;
    const int ap_lo = b.net_csr[R.net_off + a], ap_hi = b.net_csr[R.net_off + a + 1];
    const int nap = ap_hi - ap_lo;    // 1 <= nap <= XR_MAX_AP_PER_NET (checked at load)
    for (int i = tid; i < nap; i += nthr) {
        const int f = b.ap_node[R.ap_off + ap_lo + i];
        const int z = f % Z, y = (f / Z) % Y, x = f / (Y * Z);
        s_ap_l[i] = (ApIndex)lay.idx(x, y, z);
        s_ap_pin[i] = b.ap_pin[R.ap_off + ap_lo + i];
        s_ap_conn[i] = 0;
    }
    if (tid == 0) {            // layer tables: horizontal layers carry x-lines, vertical layers y-lines
        int nh = 0, nv = 0;
        for (int z = 0; z < Z; z++) {
            if ((ldir >> z) & 1u) { s_ztrk[z] = (unsigned short)(tracks_h + nv * X); s_vl[nv++] = (unsigned char)z; }
            else { s_ztrk[z] = (unsigned short)(nh * Y); s_hl[nh++] = (unsigned char)z; }
        }
        s_cnt[0][0] = s_cnt[0][1] = s_cnt[0][2] = 0;
        s_cnt[1][0] = s_cnt[1][1] = s_cnt[1][2] = 0;
    }
    __syncthreads();

    // both lines through node l become dirty (used when a node becomes a source)
    auto mark_node = [&](uint32_t* mask, int l) {
        int x, y, z;
        lay.decode(l, x, y, z);
        const bool vert = (ldir >> z) & 1u;
        // s_ztrk[z]: first track id of layer z (x-tracks 0.., y-tracks tracks_h..)
        const int tr = vert ? itV0 + (s_ztrk[z] - tracks_h + x) * chV + y / XR_CH : (s_ztrk[z] + y) * chH + x / XR_CH;
        const int cl = itC0 + (x * Y + y) * chC + (ZCH > 0 ? 0 : z / XR_CH);
        atomicOr(&mask[tr >> 5], 1u << (tr & 31));
        atomicOr(&mask[cl >> 5], 1u << (cl & 31));
    };

    // ---- component = all access points of the lowest pin id -------------------------------------
    if (tid == 0) {
        int first = 0x7FFFFFFF;
        for (int i = 0; i < nap; i++) first = min(first, (int)s_ap_pin[i]);
        int npins = 0;
        for (int i = 0; i < nap; i++) {
            bool seen = false;
            for (int j = 0; j < i; j++) if (s_ap_pin[j] == s_ap_pin[i]) { seen = true; break; }
            npins += !seen;
            if (s_ap_pin[i] == first) {
                s_ap_conn[i] = 1;
                field[s_ap_l[i]] &= 3u;
                mark_node(s_dirty0, s_ap_l[i]);
            }
        }
        s_remaining = npins - 1;
        s_bound = XR_INF;
    }
    __syncthreads();
    XR_LAP(0);

    const uint32_t via4 = (uint32_t)b.via_cost << 2;
    const uint32_t pen4 = (uint32_t)b.pen_cost << 2;
    int d_vio = 0, d_wl = 0, d_via = 0, plen = 0, status = XR_ENV_OK, nsweeps = 0;   // thread 0 only
    uint64_t h = (tid == 0) ? b.hash[e] : 0;
    int32_t* __restrict__ path = b.path + (int64_t)e * b.path_cap;
    uint32_t* cur = s_dirty0;       // dirty lines to visit this iteration (LDS variant)
    uint32_t* nxt = s_dirty1;       // dirty lines reported during this iteration
    int parity = 0;

    const int round_cap = b.round_cap > 0 ? b.round_cap : 1024 + N;     // hang guard, see xr_dial.h
    while (s_remaining > 0) {
        // new search: the bound was reset, so lines that refused candidates must be looked at again
        for (int i = tid; i < nlw; i += nthr) { const uint32_t m = s_defer[i]; if (m) { atomicOr(&cur[i], m); s_defer[i] = 0; } }
        // ---- relax to the (pruned) fixpoint ----------------------------------------------------
        bool aborted = false;
        for (int nss = 0;; nss++) {
            // bound from the targets' current distances
            for (int i = tid; i < nap; i += nthr)
                if (!s_ap_conn[i]) { const uint32_t w = field[s_ap_l[i]]; if (w < XR_W_UNREACHED) atomicMin(&s_bound, w >> 2); }
            int nH, nV, nC;
            {
                // compact the dirty bitmask into dense worklists (one list per line kind)
                for (int wi = tid; wi < nlw; wi += nthr) {
                    uint32_t m = cur[wi];
                    const int id0 = wi << 5;
                    if (m) {
                        cur[wi] = 0;
                        // per kind: reserve a run in the list with one atomic, then fill it
                        uint32_t mh = 0, mv = 0, mc = 0;
                        if (id0 + 32 <= itH) mh = m;
                        else if (id0 >= itC0) mc = m;
                        else if (id0 >= itV0 && id0 + 32 <= itC0) mv = m;
                        else {
                            for (uint32_t t = m; t; t &= t - 1) {
                                const int bit = __ffs((int)t) - 1, id = id0 + bit;
                                if (id < itH) mh |= 1u << bit; else if (id < itC0) mv |= 1u << bit; else mc |= 1u << bit;
                            }
                        }
                        if (mh) { int o = atomicAdd(&s_cnt[parity][0], __popc(mh)); for (; mh; mh &= mh - 1) s_listH[o++] = (unsigned short)(id0 + __ffs((int)mh) - 1); }
                        if (mv) { int o = atomicAdd(&s_cnt[parity][1], __popc(mv)); for (; mv; mv &= mv - 1) s_listV[o++] = (unsigned short)(id0 + __ffs((int)mv) - 1 - itV0); }
                        if (mc) { int o = atomicAdd(&s_cnt[parity][2], __popc(mc)); for (; mc; mc &= mc - 1) s_listC[o++] = (unsigned short)(id0 + __ffs((int)mc) - 1 - itC0); }
                    }
                }
                __syncthreads();
                nH = s_cnt[parity][0]; nV = s_cnt[parity][1]; nC = s_cnt[parity][2];
                if (nH + nV + nC == 0) break;              // uniform: nothing left to visit
                if (nss >= round_cap) { aborted = true; break; }      // uniform
                if (tid == 0) { s_cnt[parity ^ 1][0] = 0; s_cnt[parity ^ 1][1] = 0; s_cnt[parity ^ 1][2] = 0; }
            }
            XR_LAP(1);
#ifdef XR_PHASE_TIMING
            if (tid == 0) { _ph[6] += nH + nV + nC; _ph[7] += 1; }
#endif
            // candidates must stay <= min(search bound, XR_DIST_CAP - 1): boundw1 = (that + 1) << 2
            const uint32_t bnd = min(s_bound, (uint32_t)(XR_W_USABLE_END >> 2) - 1u);
            const uint32_t bound4 = (bnd + 1u) << 2;
            // each kind starts on a wave boundary: no divergence between line kinds inside a wave
            const int offV = (nH + 63) & ~63, offC = offV + ((nV + 63) & ~63);
            const int total = offC + nC;
            for (int k = tid; k < total; k += nthr) {
                if (k < offV) {
                    if (k < nH) {
                        const int it = (int)s_listH[k];
                        const int t = it / chH, c0 = it - t * chH;
                        const int zi = t / Y, y = t - zi * Y;
                        const int base = lay.idx(0, y, s_hl[zi]), stride = lay.LDS_DIST ? lay.SX : 1;
                        auto ix = [=](int i) { return base + i * stride; };
                        // lowered node x: column (x, y) must be looked at (the chunk of the column holding this layer)
                        const int zc = (ZCH > 0) ? 0 : (int)s_hl[zi] / XR_CH;
                        auto mk = [&](int start, uint32_t bits) {
                            while (bits) {
                                const int id = itC0 + ((start + __ffs((int)bits) - 1) * Y + y) * chC + zc; bits &= bits - 1;
                                atomicOr(&nxt[id >> 5], 1u << (id & 31));
                            }
                        };
                        auto df = [&](int c) { const int id = t * chH + c; atomicOr(&s_defer[id >> 5], 1u << (id & 31)); };
                        xr_seg_pass<true, true, XR_CH, false, !LDS_DIST>(field, s_el4x, ix, X, 0u, pen4, bound4, c0, mk, df);
                        xr_seg_pass<false, true, XR_CH, false, !LDS_DIST>(field, s_el4x, ix, X, 0u, pen4, bound4, c0, mk, df);
                    }
                } else if (k < offC) {
                    if (k - offV < nV) {
                        const int it = (int)s_listV[k - offV];
                        const int t = it / chV, c0 = it - t * chV;
                        const int zi = t / X, x = t - zi * X;
                        const int base = lay.idx(x, 0, s_vl[zi]), stride = lay.LDS_DIST ? lay.SY : 1;
                        auto ix = [=](int i) { return base + i * stride; };
                        const int zc = (ZCH > 0) ? 0 : (int)s_vl[zi] / XR_CH;
                        auto mk = [&](int start, uint32_t bits) {
                            if (chC == 1) xr_or_run(nxt, itC0 + x * Y + start, bits);
                            else while (bits) {
                                const int id = itC0 + (x * Y + start + __ffs((int)bits) - 1) * chC + zc; bits &= bits - 1;
                                atomicOr(&nxt[id >> 5], 1u << (id & 31));
                            }
                        };
                        auto df = [&](int c) { const int id = itV0 + t * chV + c; atomicOr(&s_defer[id >> 5], 1u << (id & 31)); };
                        xr_seg_pass<true, true, XR_CH, false, !LDS_DIST>(field, s_el4y, ix, Y, 0u, pen4, bound4, c0, mk, df);
                        xr_seg_pass<false, true, XR_CH, false, !LDS_DIST>(field, s_el4y, ix, Y, 0u, pen4, bound4, c0, mk, df);
                    }
                } else {
                    const int it = (int)s_listC[k - offC];
                    const int c = it / chC, c0 = it - c * chC;
                    const int x = c / Y, y = c - x * Y;
                    // lowered node z: the chunk of its track that holds this column must be looked at
                    auto mk = [&](int start, uint32_t bits) {
                        while (bits) {
                            const int z = start + __ffs((int)bits) - 1; bits &= bits - 1;
                            const bool vert = (ldir >> z) & 1u;
                            const int id = vert ? itV0 + (s_ztrk[z] - tracks_h + x) * chV + y / XR_CH
                                                : (s_ztrk[z] + y) * chH + x / XR_CH;
                            atomicOr(&nxt[id >> 5], 1u << (id & 31));
                        }
                    };
                    auto df = [&](int cc) { const int id = itC0 + c * chC + cc; atomicOr(&s_defer[id >> 5], 1u << (id & 31)); };
                    // via chain: unit stride in the packed layout; one word per layer plane in the layer-major layout
                    const bool packed = lay.LDS_DIST;
                    const int cbase = packed ? lay.idx(x, y, 0) : 0, offh = y * X + x, offv = x * Y + y, xy = lay.XY;
                    auto ix = [=](int z) { return packed ? cbase + z : z * xy + (((ldir >> z) & 1u) ? offv : offh); };
                    if (ZCH > 0) {
                        xr_seg_pass<true, false, (ZCH > 0 ? ZCH : 1), true, !LDS_DIST>(field, nullptr, ix, ZCH, via4, pen4, bound4, 0, mk, df);
                        xr_seg_pass<false, false, (ZCH > 0 ? ZCH : 1), true, !LDS_DIST>(field, nullptr, ix, ZCH, via4, pen4, bound4, 0, mk, df);
                    } else {
                        xr_seg_pass<true, false, XR_CH, false, !LDS_DIST>(field, nullptr, ix, Z, via4, pen4, bound4, c0, mk, df);
                        xr_seg_pass<false, false, XR_CH, false, !LDS_DIST>(field, nullptr, ix, Z, via4, pen4, bound4, c0, mk, df);
                    }
                }
            }
            nsweeps++;
            __syncthreads();        // (workgroup-scope: also orders the HBM-scratch field of the large-region variant)
            uint32_t* t = cur; cur = nxt; nxt = t;
            parity ^= 1;
            XR_LAP(2);
        }

        if (aborted) {                        // uniform: the remaining pins are charged as unreachable, nothing is traced
            if (tid == 0) { d_vio += s_remaining; status |= XR_ENV_ROUTER_ABORT | XR_ENV_UNREACHABLE; s_remaining = 0; s_target_i = -1; }
        } else
        // ---- nearest access point of an unconnected pin; ties -> lowest flat index (wave 0) --------
        if (tid < 64) {
            if (LDS_DIST) for (int i = tid; i < claim_words; i += 64) s_claim[i] = 0;   // aliases the (dead) worklists
            unsigned long long best = ~0ULL;
            for (int i = tid; i < nap; i += 64) {
                if (s_ap_conn[i]) continue;
                const uint32_t w = field[s_ap_l[i]];
                if (w >= XR_W_UNREACHED) continue;
                // (distance, flat node index): ties go to the lowest FLAT index f = (x*Y+y)*Z+z, whatever the
                // field layout is
                int ax, ay, az;
                lay.decode(s_ap_l[i], ax, ay, az);
                const unsigned long long key = ((unsigned long long)(w >> 2) << 32) | (unsigned)((ax * Y + ay) * Z + az);
                best = key < best ? key : best;
            }
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long o = __shfl_xor(best, off);
                best = o < best ? o : best;
            }
            int best_i = -1;
            if (best != ~0ULL) {                  // AP slot holding that node (node ids are unique per net)
                const int bf = (int)(best & 0xFFFFFFFFu);
                const int best_l = lay.idx(bf / (Y * Z), (bf / Z) % Y, bf % Z);
                for (int i0 = 0; i0 < nap && best_i < 0; i0 += 64) {
                    const int i = i0 + tid;
                    const unsigned long long m = __ballot(i < nap && (int)s_ap_l[i] == best_l);
                    if (m) best_i = i0 + __ffsll((long long)m) - 1;
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
                // ---- deterministic back-trace: first predecessor in the order E,S,W,N,U,D.
                // Lanes 0..5 test one direction each; ballot + ffs picks the first match.  The field is
                // only READ here; claimed nodes are zeroed afterwards (a zero written mid-trace could
                // fake a predecessor match on the node after it).
                int v = s_ap_l[best_i];
                uint32_t vw = field[v];
                for (int nt = 0; (vw >> 2) > 0; nt++) {
                    if (nt > N) { if (tid == 0) status |= 0x100; break; }      // (hang guard: distances strictly decrease)
                    int x, y, z;
                    lay.decode(v, x, y, z);
                    const uint32_t need4 = (vw & ~3u) - ((vw & 2u) ? pen4 : 0u);   // pred distance + edge, x4
                    const bool vert = (ldir >> z) & 1u;
                    int u = -1;
                    uint32_t len4 = 0;
                    switch (tid) {
                    case 0: if (!vert && x + 1 < X) { u = lay.idx(x + 1, y, z); len4 = s_el4x[x + 1]; } break;   // E
                    case 1: if (vert && y > 0)      { u = lay.idx(x, y - 1, z); len4 = s_el4y[y]; } break;       // S
                    case 2: if (!vert && x > 0)     { u = lay.idx(x - 1, y, z); len4 = s_el4x[x]; } break;       // W
                    case 3: if (vert && y + 1 < Y)  { u = lay.idx(x, y + 1, z); len4 = s_el4y[y + 1]; } break;   // N
                    case 4: if (z + 1 < Z)          { u = lay.idx(x, y, z + 1); len4 = via4; } break;           // U
                    case 5: if (z > 0)              { u = lay.idx(x, y, z - 1); len4 = via4; } break;           // D
                    default: break;
                    }
                    uint32_t uw = XR_W_BLOCK;
                    bool ok = false;
                    if (u >= 0) {
                        uw = field[u];
                        ok = (uw - 1u) < (XR_W_USABLE_END - 1u) && (uw & ~3u) + len4 == need4;
                    }
                    const unsigned long long m = __ballot(ok);
                    if (m == 0) { if (tid == 0) status |= 0x100; break; }     // inconsistent field: cannot happen
                    const int src = __ffsll((long long)m) - 1;
                    const int pu = __shfl(u, src);
                    const uint32_t puw = __shfl(uw, src), pl4 = __shfl(len4, src);
                    if (tid == 0) {                 // claim v
                        const int f = (x * Y + y) * Z + z;
                        if (vw & 2u) d_vio += 1;
                        s_claim[v >> 5] |= 1u << (v & 31);
                        if (plen < b.path_cap) path[plen] = f;
                        plen++;
                        fnv_mix(h, (uint32_t)f);
                        if (src >= 4) d_via += 1; else d_wl += (int)(pl4 >> 2);
                    }
                    v = pu; vw = puw;
                }
                if (tid == 0 && (status & 0x100)) {
                    s_remaining = 0;              // never taken on a consistent field; avoids spinning
                } else if (tid == 0) {
                    // terminal node of the component: claimed (and recorded) only if nobody holds it yet
                    int x, y, z;
                    lay.decode(v, x, y, z);
                    const int f = (x * Y + y) * Z + z;
                    if (owner[f] == 0) {
                        owner[f] = (int16_t)a;
                        if (plen < b.path_cap) path[plen] = f;
                        plen++;
                        fnv_mix(h, (uint32_t)f);
                    }
                    s_remaining -= 1;
                    s_bound = XR_INF;
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
                    if (s_ap_pin[i] == pin) {
                        s_ap_conn[i] = 1;
                        field[s_ap_l[i]] &= 3u;
                        mark_node(cur, s_ap_l[i]);
                    }
                for (int wi = tid; wi < claim_words; wi += nthr) {
                    uint32_t m = s_claim[wi];
                    if (m) {
                        s_claim[wi] = 0;
                        while (m) {
                            const int l = (wi << 5) + __ffs((int)m) - 1;
                            m &= m - 1;
                            field[l] &= 3u;
                            mark_node(cur, l);
                            // claim the path node if nobody holds it (done here, by many threads at once, instead
                            // of one dependent HBM load per node inside the serial back-trace)
                            int cx, cy, cz;
                            lay.decode(l, cx, cy, cz);
                            const int cf = (cx * Y + cy) * Z + cz;
                            if (owner[cf] == 0) owner[cf] = (int16_t)a;
                        }
                    }
                }
            }
        }
        __syncthreads();
        XR_LAP(4);
    }

    // ---- Game.step bookkeeping (reference baseline/baseline_utils.py:412, :426-438) + reward ------
    if (tid == 0) {
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
        b.sweeps[e] = nsweeps;
        fnv_mix(h, (uint32_t)a);
        fnv_mix(h, (uint32_t)d_vio); fnv_mix(h, (uint32_t)d_wl); fnv_mix(h, (uint32_t)d_via);
        fnv_mix(h, (uint32_t)plen);
        b.hash[e] = h;
        b.env_steps[e] += 1;
        atomicAdd(b.total_steps, 1ULL);
        xr_publish_record(b, e);
    }
    XR_LAP(5);
    XR_TDUMP();
}

#include "xr_dial.h"
#include "xr_dial3.h"

// router selection of the step kernels: ZCH == XR_ZCH_DIAL -> the bucketed-frontier router (xr_dial.h, the default),
// else the line-segment sweeps above (xr_config.router = XR_ROUTER_SWEEP)
#define XR_ZCH_DIAL (-1)
#define XR_ZCH_DIAL2 (-2)      // the frontier router (either form) with the XR-Maze v2 knobs compiled in
#define XR_ZCH_DIAL3 (-3)      // round 3's LDS form (xr_dial3.h: mask rounds, quads of lanes per chain, predecessor directions in the field word)
#define XR_ZCH_DIAL3V2 (-4)    // ... with the XR-Maze v2 knobs
template <bool LDS_DIST, int ZCH>
__device__ __forceinline__ void xr_route_dispatch(const XrBatchDev& b, const int e, const int a, char* smem) {
    if constexpr (ZCH == XR_ZCH_DIAL3V2) {
        static_assert(LDS_DIST, "xr_dial3.h is an LDS form");
        xr_dial3_route_env<true>(b, e, a, smem);
    } else if constexpr (ZCH == XR_ZCH_DIAL3) {
        static_assert(LDS_DIST, "xr_dial3.h is an LDS form");
        xr_dial3_route_env<false>(b, e, a, smem);
    } else if constexpr (ZCH == XR_ZCH_DIAL2) {
        if constexpr (LDS_DIST) xr_dial_route_env<true>(b, e, a, smem);
        else xr_dial_route_env_big<true>(b, e, a, smem);
    } else if constexpr (ZCH == XR_ZCH_DIAL) {
        if constexpr (LDS_DIST) xr_dial_route_env<false>(b, e, a, smem);
        else {
            // regions too large for LDS (round 4): first the LDS router inside a window around the net, with an exactness certificate
            // (xr_dial3.h, WIN); the HBM-scratch form only for nets that do not fit the window or whose certificate fails
            if (b.win_x > 0 && xr_dial3_route_env<false, true>(b, e, a, smem)) return;
            xr_dial_route_env_big<false>(b, e, a, smem);
        }
    } else {
        xr_route_env<LDS_DIST, ZCH>(b, e, a, smem);
    }
}

// ------------------------------------------------------------------------------------------------
// random net-order policy: j-th legal net, j from a counter-based hash
// ------------------------------------------------------------------------------------------------
__global__ void xr_random_action_kernel(XrBatchDev b, int32_t* __restrict__ actions, uint64_t seed) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= b.n_envs) return;
    const int nl = b.nlegal[e];
    if (nl == 0) { actions[e] = 0; return; }
    const uint64_t r = splitmix64(seed ^ splitmix64((uint64_t)e * 0x100000001B3ULL + (uint64_t)b.env_steps[e]));
    int j = (int)(r % (uint64_t)nl);
    const uint64_t* lw = b.legal + (int64_t)e * b.legal_words;
    int act = 0;
    for (int w = 0; w < b.legal_words; w++) {
        uint64_t m = lw[w];
        const int pc = __popcll(m);
        if (j >= pc) { j -= pc; continue; }
        while (j > 0) { m &= m - 1; j--; }
        act = w * 64 + __ffsll((long long)m);     // 1-based bit position == 1-based net id
        break;
    }
    actions[e] = act;
}

// ------------------------------------------------------------------------------------------------
// observation (build_3Dgrid, reference baseline/build_3Dgrid.py:94-188,224-270)
//
//   ch 0           obstacle: Net == -1, or is_used (NORMAL or ACCESS)                    (:19-36,:94-103)
//   ch 1           net ids of netSet (ascending) at flat positions 0..K-1                (:144-161)
//   ch 2+7i        access points of the i-th net, any pin                                 (:111-120)
//   ch 2+7i+1..6   ONE aliased tensor (:125): AP that has an in-bounds axis neighbour which is an AP
//                  of the same net (:126-138)
// The channel-plane index of vertex (x,y,z) is (x*Y+y)*Z+z: reshape without permute (:103,:142).
//
// "Is an AP of net n with such a neighbour" does not depend on which channel is being written, so a
// thread decides it once for its 4 consecutive nodes and then streams K*7+2 float4 stores, each wave
// store covering 1 KiB of one channel plane.  HBM-write bound: 4*N*(2+7K) bytes per env.
// ------------------------------------------------------------------------------------------------
struct XrStateSrc {          // compact batch state
    const int16_t* node_net;
    const int16_t* owner;
    __device__ __forceinline__ int net(int f) const { const int n = node_net[f]; return n; }
    __device__ __forceinline__ bool used(int f) const { return owner[f] != 0; }
};
struct XrRecSrc {            // packed records (stateless entry point)
    const uint32_t* rec;
    __device__ __forceinline__ int net(int f) const {
        const uint32_t r = rec[f];
        const uint32_t t = XR_REC_TYPE(r);
        return t == XR_TYPE_ACCESS ? (int)XR_REC_NET1(r) : (t == XR_TYPE_BLOCKAGE ? -1 : 0);
    }
    __device__ __forceinline__ bool used(int f) const { return XR_REC_USED(rec[f]) != 0; }
};

template <class Src>
__device__ __forceinline__ void xr_node_features(const Src& s, int f, int X, int Y, int Z, int N, float& obst,
                                                 int& apnet, bool& adj) {
    if (f >= N) { obst = 0.f; apnet = 0; adj = false; return; }
    const int n = s.net(f);
    const bool used = s.used(f);           // (unconditional: the loads of an unrolled caller are issued together)
    obst = (n == -1 || used) ? 1.f : 0.f;
    apnet = n > 0 ? n : 0;
    adj = false;
    if (n > 0) {
        const int z = f % Z, y = (f / Z) % Y, x = f / (Y * Z);
        const int YZ = Y * Z;
        if (x + 1 < X && s.net(f + YZ) == n) adj = true;
        else if (y > 0 && s.net(f - Z) == n) adj = true;
        else if (x > 0 && s.net(f - YZ) == n) adj = true;
        else if (y + 1 < Y && s.net(f + Z) == n) adj = true;
        else if (z + 1 < Z && s.net(f + 1) == n) adj = true;
        else if (z > 0 && s.net(f - 1) == n) adj = true;
    }
}

// VEC = 4: float4 stores (needs N % 4 == 0 and 16-byte aligned planes); VEC = 1: any N.
#ifndef XR_OBS_PLAIN_STORES      // non-temporal: the observation is written once and read by somebody else
typedef float xr_f4 __attribute__((ext_vector_type(4)));
#define XR_ST4(ptr, val) __builtin_nontemporal_store(xr_f4{(val).x, (val).y, (val).z, (val).w}, reinterpret_cast<xr_f4*>(ptr))
#else
#define XR_ST4(ptr, val) (*reinterpret_cast<float4*>(ptr) = (val))
#endif
template <class Src, int VEC>
__device__ __forceinline__ void xr_obs_write(const Src& s, int X, int Y, int Z, int N, const int* s_ids, int K,
                                             float* __restrict__ out, int chunk_base, int knets = -1) {
    if (knets < 0) knets = K;            // net plane groups written here (0: planes 0..1 only)
    const int f0 = chunk_base + threadIdx.x * VEC;
    if (f0 >= N) return;
    float obst[VEC];
    int apnet[VEC];
    bool adj[VEC];
#pragma unroll
    for (int j = 0; j < VEC; j++) xr_node_features(s, f0 + j, X, Y, Z, N, obst[j], apnet[j], adj[j]);
    float* p = out + f0;
    if (VEC == 4) {
        float4 v;
        v.x = obst[0]; v.y = obst[1]; v.z = obst[2]; v.w = obst[3];
        XR_ST4(p, v);
        p += N;
        v.x = (f0 + 0 < K) ? (float)s_ids[f0 + 0] : 0.f;
        v.y = (f0 + 1 < K) ? (float)s_ids[f0 + 1] : 0.f;
        v.z = (f0 + 2 < K) ? (float)s_ids[f0 + 2] : 0.f;
        v.w = (f0 + 3 < K) ? (float)s_ids[f0 + 3] : 0.f;
        XR_ST4(p, v);
        p += N;
        bool anyap = false;
#pragma unroll
        for (int j = 0; j < VEC; j++) anyap |= (apnet[j] != 0);
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = 0; i < knets; i++) {
            float4 m = zero, ma = zero;
            if (anyap) {
                const int id = s_ids[i];
                m.x = (apnet[0] == id) ? 1.f : 0.f; m.y = (apnet[1] == id) ? 1.f : 0.f;
                m.z = (apnet[2] == id) ? 1.f : 0.f; m.w = (apnet[3] == id) ? 1.f : 0.f;
                ma.x = (adj[0] ? m.x : 0.f); ma.y = (adj[1] ? m.y : 0.f);
                ma.z = (adj[2] ? m.z : 0.f); ma.w = (adj[3] ? m.w : 0.f);
            }
            XR_ST4(p, m);
            p += N;
#pragma unroll
            for (int c = 0; c < 6; c++) { XR_ST4(p, ma); p += N; }
        }
    } else {
        p[0] = obst[0];
        p += N;
        p[0] = (f0 < K) ? (float)s_ids[f0] : 0.f;
        p += N;
        for (int i = 0; i < knets; i++) {
            const float m = (apnet[0] != 0 && apnet[0] == s_ids[i]) ? 1.f : 0.f;
            const float ma = adj[0] ? m : 0.f;
            p[0] = m;
            p += N;
#pragma unroll
            for (int c = 0; c < 6; c++) { p[0] = ma; p += N; }
        }
    }
}

// float4 stores for ANY N (design-derived regions rarely have N % 4 == 0): the env's (2+7K)*N floats are ONE
// contiguous run starting at a 16-byte aligned address, so it is written as a flat stream of aligned float4
// slots, whatever N is.  The workgroup first reduces every node to a 16-bit feature in LDS
//     AP net id (14 bits) | has-same-net-axis-neighbour << 14 | obstacle << 15
// and then every thread walks slots tid, tid+T, ...; the (plane, node) of a slot advances incrementally and a
// slot that straddles two planes is resolved float by float.
__device__ __forceinline__ float xr_plane_value(unsigned ft, int plane, int node, const int* s_ids, int K) {
    if (plane >= 2) {
        const int pi = plane - 2, i = pi / 7, c = pi - 7 * i;
        const unsigned id = (unsigned)s_ids[i];
        return c ? (((ft & 0x7FFFu) == (id | 0x4000u)) ? 1.f : 0.f) : (((ft & 0x3FFFu) == id) ? 1.f : 0.f);
    }
    if (plane == 0) return (ft & 0x8000u) ? 1.f : 0.f;
    return node < K ? (float)s_ids[node] : 0.f;
}

template <class Src>
__device__ __forceinline__ void xr_obs_env_stream(const Src& src, int X, int Y, int Z, int N, const int* s_ids, int K,
                                                  float* __restrict__ out, unsigned short* s_feat, int planes = -1) {
    const int tid = threadIdx.x, nthr = blockDim.x;
#pragma unroll 4
    for (int f = tid; f < N; f += nthr) {
        float obst; int apnet; bool adj;
        xr_node_features(src, f, X, Y, Z, N, obst, apnet, adj);
        s_feat[f] = (unsigned short)(apnet | (adj ? 0x4000 : 0) | (obst != 0.f ? 0x8000 : 0));
    }
    __syncthreads();
    // `planes` >= 0: only the first `planes` planes are written here (split form: the rest comes from the writer kernel)
    const long long total = (long long)(planes >= 0 ? planes : 2 + 7 * K) * N;          // floats
    const long long nslot = total >> 2;
    int plane = (int)((4LL * tid) / N), node = (int)((4LL * tid) - (long long)plane * N);
    const int dplane = (4 * nthr) / N, dnode = (4 * nthr) - dplane * N;
    for (long long sidx = tid; sidx < nslot; sidx += nthr) {
        float v[4];
        if (node + 3 < N && plane >= 2) {            // common case: four nodes of one net plane
            const int pi = plane - 2, i = pi / 7, c = pi - 7 * i;
            const unsigned id = (unsigned)s_ids[i];
            const unsigned msk = c ? 0x7FFFu : 0x3FFFu, want = c ? (id | 0x4000u) : id;
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = ((s_feat[node + j] & msk) == want) ? 1.f : 0.f;
        } else {
            int p = plane, f = node;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                v[j] = xr_plane_value(s_feat[f], p, f, s_ids, K);
                if (++f == N) { f = 0; ++p; }
            }
        }
        XR_ST4(out + 4 * sidx, make_float4(v[0], v[1], v[2], v[3]));
        plane += dplane; node += dnode;
        if (node >= N) { node -= N; ++plane; }
    }
    // the last total % 4 floats
    const int tail = (int)(total & 3);
    if (tid < tail) {
        const long long g = (nslot << 2) + tid;
        const int p = (int)(g / N), f = (int)(g - (long long)p * N);
        out[g] = xr_plane_value(s_feat[f], p, f, s_ids, K);
    }
}

// legal bitmask -> ascending id list in LDS (== sorted(list(netSet)), build_3Dgrid.py:177)
// (the same with the thread's legal word already in a register — thread t holds word t, words <= blockDim.x — and barriers that order
//  LDS only: nothing here waits for global stores in flight)
__device__ __forceinline__ int xr_legal_ids_pre(uint64_t my_m, int words, int* s_ids, int* s_pref) {
    const int tid = threadIdx.x;
    if (tid < words) s_pref[tid] = __popcll(my_m);
    xr_lds_barrier();
    if (tid == 0) {
        int acc = 0;
        for (int w = 0; w < words; w++) { const int c = s_pref[w]; s_pref[w] = acc; acc += c; }
        s_pref[words] = acc;
    }
    xr_lds_barrier();
    if (tid < words) {
        uint64_t m = my_m;
        int o = s_pref[tid];
        while (m) { const int bit = __ffsll((long long)m) - 1; s_ids[o++] = tid * 64 + bit + 1; m &= m - 1; }
    }
    xr_lds_barrier();
    return s_pref[words];
}
__device__ __forceinline__ int xr_legal_ids(const uint64_t* __restrict__ lw, int words, int* s_ids, int* s_pref) {
    const int tid = threadIdx.x;
    for (int w = tid; w < words; w += blockDim.x) s_pref[w] = __popcll(lw[w]);
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int w = 0; w < words; w++) { const int c = s_pref[w]; s_pref[w] = acc; acc += c; }
        s_pref[words] = acc;
    }
    __syncthreads();
    for (int w = tid; w < words; w += blockDim.x) {
        uint64_t m = lw[w];
        int o = s_pref[w];
        while (m) { const int bit = __ffsll((long long)m) - 1; s_ids[o++] = w * 64 + bit + 1; m &= m - 1; }
    }
    __syncthreads();
    return s_pref[words];
}

template <int VEC>
__global__ void xr_obs_kernel(XrBatchDev b, float* __restrict__ out, int64_t env_stride, int env_lo) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* s_ids = reinterpret_cast<int*>(smem);                 // [legal_words*64]
    int* s_pref = s_ids + b.legal_words * 64;                  // [legal_words+1]
    const int e = env_lo + blockIdx.y;
    const XrRegionDev R = b.regions[b.env_region[e]];
    const int chunk_base = blockIdx.x * blockDim.x * VEC;
    if (chunk_base >= R.N) return;
    const int K = xr_legal_ids(b.legal + (int64_t)e * b.legal_words, b.legal_words, s_ids, s_pref);
    XrStateSrc src{b.rg_node_net + R.node_off, b.owner + (int64_t)e * b.n_max};
    xr_obs_write<XrStateSrc, VEC>(src, R.X, R.Y, R.Z, R.N, s_ids, K, out + (int64_t)blockIdx.y * env_stride,
                                  chunk_base);
}

// stand-alone observation as a flat float4 stream, one workgroup per env (any N; env_stride % 4 == 0, aligned base)
__global__ void xr_obs_stream_kernel(XrBatchDev b, float* __restrict__ out, int64_t env_stride, int env_lo) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* s_ids = reinterpret_cast<int*>(smem);
    int* s_pref = s_ids + b.legal_words * 64;
    unsigned short* s_feat = reinterpret_cast<unsigned short*>(s_pref + ((b.legal_words + 1 + 3) & ~3));
    const int e = env_lo + blockIdx.x;
    const XrRegionDev R = b.regions[b.env_region[e]];
    const int K = xr_legal_ids(b.legal + (int64_t)e * b.legal_words, b.legal_words, s_ids, s_pref);
    XrStateSrc src{b.rg_node_net + R.node_off, b.owner + (int64_t)e * b.n_max};
    xr_obs_env_stream(src, R.X, R.Y, R.Z, R.N, s_ids, K, out + (int64_t)blockIdx.x * env_stride, s_feat);
}

template <int VEC>
__global__ void xr_obs_records_kernel(const uint32_t* __restrict__ rec, int X, int Y, int Z,
                                      const int32_t* __restrict__ nets, int K, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* s_ids = reinterpret_cast<int*>(smem);
    for (int i = threadIdx.x; i < K; i += blockDim.x) s_ids[i] = nets[i];
    __syncthreads();
    XrRecSrc src{rec};
    xr_obs_write<XrRecSrc, VEC>(src, X, Y, Z, X * Y * Z, s_ids, K, out, blockIdx.x * blockDim.x * VEC);
}

// observation of env e's CURRENT state by the calling workgroup (the step kernel's epilogue).  head_only: the planes
// the split / queue forms leave to the step kernel (planes 0..1 + the net planes of the lowest XR_SPLIT_KEEP ranks).
__device__ __forceinline__ void xr_obs_epilogue(const XrBatchDev& b, int e, char* smem, bool head_only) {
    __syncthreads();          // this workgroup's owner / legal / region writes are visible to all its threads
    const XrRegionDev R = b.regions[b.env_region[e]];
    int* s_ids = reinterpret_cast<int*>(smem);              // the field is dead: reuse its LDS
    int* s_pref = s_ids + b.legal_words * 64;
    XrStateSrc src{b.rg_node_net + R.node_off, b.owner + (int64_t)e * b.n_max};
    float* __restrict__ out = b.obs_out + (int64_t)e * b.obs_stride;
    // Aligned planes, and the env's two state rows fit the launch's LDS beside the id list: the rows (node_net, owner: 2 bytes per node
    // each) first go to LDS with 16-byte loads that are all in flight together with the legal words, and the features come from there.
    // Read straight from global memory, every 1024-node chunk is its own chain of dependent loads (node -> up to six neighbours)
    // before its stores — ~9 chains per env, each several microseconds while the unit writers of the other workgroups saturate the
    // memory pipeline (tools/queue_timeline_probe.py: 85 us per env at 512 envs, more than the route before it).
    const int ids_bytes = ((b.legal_words * 64 + b.legal_words + 1) * 4 + 15) & ~15;
    const int npad = (R.N + 7) & ~7;
    if (b.obs_vec4 == 1 && ids_bytes + 4 * npad <= b.obs_lds_bytes && b.legal_words <= (int)blockDim.x) {
        const int tid = threadIdx.x, nthr = (int)blockDim.x;
        const uint64_t my_m = tid < b.legal_words ? b.legal[(int64_t)e * b.legal_words + tid] : 0ull;
        int16_t* l_nn = reinterpret_cast<int16_t*>(smem + ids_bytes);
        int16_t* l_ow = l_nn + npad;
        const int nchunk = npad >> 3;
        // (unrolled by four: the scheduler issues the eight loads of a group together; register arrays here ended up in scratch)
#pragma unroll 4
        for (int ci = tid; ci < nchunk; ci += nthr) {
            const int4 vn = *reinterpret_cast<const int4*>(src.node_net + (ci << 3));
            const int4 vo = *reinterpret_cast<const int4*>(src.owner + (ci << 3));
            *reinterpret_cast<int4*>(l_nn + (ci << 3)) = vn;
            *reinterpret_cast<int4*>(l_ow + (ci << 3)) = vo;
        }
        const int K = xr_legal_ids_pre(my_m, b.legal_words, s_ids, s_pref);        // (its barriers also publish the rows)
        const int knets = head_only ? XR_SPLIT_KEEP(b, K) : K;
        XrStateSrc lsrc{l_nn, l_ow};
        for (int cb = 0; cb < R.N; cb += nthr * 4)
            xr_obs_write<XrStateSrc, 4>(lsrc, R.X, R.Y, R.Z, R.N, s_ids, K, out, cb, knets);
        return;
    }
    // Unaligned planes (stream form): the feature pass reads node_net of every node and of up to six neighbours — that row goes to LDS
    // the same way (behind the 2-byte feature list); `owner` is read once per node, straight from global memory.
    const int feat_bytes = (b.legal_words * 64 + ((b.legal_words + 1 + 3) & ~3)) * 4;
    if (b.obs_vec4 == 2 && feat_bytes + 4 * npad <= b.obs_lds_bytes && b.legal_words <= (int)blockDim.x) {
        const int tid = threadIdx.x, nthr = (int)blockDim.x;
        const uint64_t my_m = tid < b.legal_words ? b.legal[(int64_t)e * b.legal_words + tid] : 0ull;
        unsigned short* s_feat = reinterpret_cast<unsigned short*>(smem + feat_bytes);
        int16_t* l_nn = reinterpret_cast<int16_t*>(smem + feat_bytes + 2 * npad);
        const int nchunk = npad >> 3;
#pragma unroll 4
        for (int ci = tid; ci < nchunk; ci += nthr)
            *reinterpret_cast<int4*>(l_nn + (ci << 3)) = *reinterpret_cast<const int4*>(src.node_net + (ci << 3));
        const int K = xr_legal_ids_pre(my_m, b.legal_words, s_ids, s_pref);        // (its barriers also publish the row)
        XrStateSrc lsrc{l_nn, src.owner};
        const int planes = head_only ? 2 + 7 * XR_SPLIT_KEEP(b, K) : -1;
        xr_obs_env_stream(lsrc, R.X, R.Y, R.Z, R.N, s_ids, K, out, s_feat, planes);
        return;
    }
    const int K = xr_legal_ids(b.legal + (int64_t)e * b.legal_words, b.legal_words, s_ids, s_pref);
    if (b.obs_vec4 == 1) {
        const int knets = head_only ? XR_SPLIT_KEEP(b, K) : K;
        for (int cb = 0; cb < R.N; cb += (int)blockDim.x * 4)
            xr_obs_write<XrStateSrc, 4>(src, R.X, R.Y, R.Z, R.N, s_ids, K, out, cb, knets);
    } else if (b.obs_vec4 == 2) {
        unsigned short* s_feat = reinterpret_cast<unsigned short*>(s_pref + ((b.legal_words + 1 + 3) & ~3));
        const int planes = head_only ? 2 + 7 * XR_SPLIT_KEEP(b, K) : -1;
        xr_obs_env_stream(src, R.X, R.Y, R.Z, R.N, s_ids, K, out, s_feat, planes);
    } else {
        for (int cb = 0; cb < R.N; cb += (int)blockDim.x)
            xr_obs_write<XrStateSrc, 1>(src, R.X, R.Y, R.Z, R.N, s_ids, K, out, cb);
    }
}

// ------------------------------------------------------------------------------------------------
// Launch order of a route-only launch.  A batch with more env slots than the chip holds workgroups runs in several
// rounds, and the launch ends with whichever long route started last: routes take 0.3x .. 6x the mean (the time follows the
// chosen net's extent and pin count), so handing the slots out longest-predicted-first shortens the launch by about a third
// (tools/lpt_probe.py: 0.68 of the slot-order makespan at 4096 ispd18_test1-sized envs; then bound by the single longest route).
// One workgroup: counting sort of the env slots by the static work class of the chosen net (descending; 0 = nothing to
// route: action out of range / not legal / env done).  The order never affects results — envs are independent.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int xr_route_work_class(const XrBatchDev& b, const int32_t* __restrict__ actions, int e) {
    const int a = actions[e];
    const XrRegionDev& R = b.regions[b.env_region[e]];
    if (a < 1 || a > R.n_nets || b.done[e]) return 0;
    if (!((b.legal[(int64_t)e * b.legal_words + ((a - 1) >> 6)] >> ((a - 1) & 63)) & 1ull)) return 0;
    if (b.net_meas) { const int m = b.net_meas[R.net_off + a]; if (m) return m; }          // measured the last time this (region, net) was routed
    return b.net_work[R.net_off + a];                                                          // the geometric guess (extent x (6 + pins))
}

__global__ void __launch_bounds__(1024) xr_route_order_kernel(XrBatchDev b, const int32_t* __restrict__ actions, int32_t* __restrict__ order) {
    __shared__ uint32_t s_cnt[256];
    __shared__ uint32_t s_wsum[4];
    const int t = threadIdx.x;
    if (t < 256) s_cnt[t] = 0;
    __syncthreads();
    int cls[8];                                      // classes of this thread's first 8 slots stay in registers (batches <= 8192)
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int e = t + k * 1024;
        cls[k] = e < b.n_envs ? xr_route_work_class(b, actions, e) : -1;
        if (cls[k] >= 0) atomicAdd(&s_cnt[cls[k]], 1u);
    }
    for (int e = t + 8 * 1024; e < b.n_envs; e += 1024) atomicAdd(&s_cnt[xr_route_work_class(b, actions, e)], 1u);
    __syncthreads();
    // exclusive prefix over DESCENDING class: thread t owns class 255 - t
    uint32_t v = 0, incl = 0;
    if (t < 256) {
        v = s_cnt[255 - t];
        incl = v;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = __shfl_up(incl, o, 64);
            if ((t & 63) >= o) incl += u;
        }
        if ((t & 63) == 63) s_wsum[t >> 6] = incl;
    }
    __syncthreads();
    if (t < 256) {
        uint32_t base = 0;
        for (int w = 0; w < (t >> 6); w++) base += s_wsum[w];
        s_cnt[255 - t] = base + incl - v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (cls[k] >= 0) order[atomicAdd(&s_cnt[cls[k]], 1u)] = t + k * 1024;
    for (int e = t + 8 * 1024; e < b.n_envs; e += 1024) order[atomicAdd(&s_cnt[xr_route_work_class(b, actions, e)], 1u)] = e;
}

// ------------------------------------------------------------------------------------------------
// The step kernel: route (xr_route_env) and, when the caller asked for it (xr_batch_step_observe), the
// observation of the new state written by the same workgroup.  Fusing the two lets the HBM-write-bound
// observation stream of some workgroups overlap the latency-bound routing of others on the same CU.
// ------------------------------------------------------------------------------------------------
template <bool LDS_DIST, int ZCH>
__global__ void __launch_bounds__(1024, 4) xr_route_kernel(XrBatchDev b, const int32_t* __restrict__ actions) {     // (register budget of 4 waves per SIMD: its own 4 workgroups per CU; without it the compiler aims higher and spills to scratch)
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef XR_TIMELINE
    // `make timeline`: absolute 100 MHz timestamps of this workgroup (start, routed, written) + where it ran
    if (threadIdx.x == 0) {
        b.phase_cycles[(int64_t)blockIdx.x * 8 + 0] = (long long)wall_clock64();
        b.phase_cycles[(int64_t)blockIdx.x * 8 + 3] = (long long)__smid();
    }
#endif
    // (env_base > 0: one launch per region, stream-per-region mode; route_order: longest predicted route first)
    const int e_ = b.route_order ? b.route_order[blockIdx.x] : (int)blockIdx.x + b.env_base;
    xr_route_dispatch<LDS_DIST, ZCH>(b, e_, actions[e_], smem);
#ifdef XR_TIMELINE
    if (threadIdx.x == 0) b.phase_cycles[(int64_t)blockIdx.x * 8 + 1] = (long long)wall_clock64();
#endif
    if (b.obs_out) xr_obs_epilogue(b, e_, smem, b.obs_head_only != 0);
#ifdef XR_TIMELINE
    __syncthreads();
    if (threadIdx.x == 0) b.phase_cycles[(int64_t)blockIdx.x * 8 + 2] = (long long)wall_clock64();
#endif
}

// ------------------------------------------------------------------------------------------------
// split observation.  7K of the 2+7K planes of an env do not depend on the routing result at all: the planes of
// net n are functions of the region's static node_net array, and WHICH nets remain after the step follows from
// the state before it (legal set, action validity, auto-reset + region rotation).  xr_plan_kernel derives that
// post-step state for every env and lists one unit per (env, remaining net); xr_netplane_kernel then streams the
// 7 planes of every unit, concurrently with the route kernel (which writes planes 0..1 in its epilogue).
// Units are equal-sized and visited in address order by the whole grid, so the write stream is balanced and
// globally sequential (measured: the fastest write pattern on this device, tools/micro/write_bw.hip).
// ------------------------------------------------------------------------------------------------
// BT threads per workgroup.  `next_queue`: the counters of the NEXT call (the two banks alternate, xr_batch.cpp) are zeroed here, so no
// call needs a memset of its own.  `order` (single-workgroup launches only, n_envs <= BT): the route tasks' longest-predicted-first
// order of xr_route_order_kernel, computed by the same launch — a small batch pays for every dependent dispatch (~8 us each).
template <int BT>
__global__ void __launch_bounds__(BT) xr_plan_kernel(XrBatchDev b, const int32_t* __restrict__ actions, uint32_t* __restrict__ next_queue,
                                                     int32_t* __restrict__ order) {
    // one env per thread; the units of a workgroup's BT envs are contiguous and in env order (block scan), the
    // workgroups reserve their runs with one atomic each (queue[2], zero when this launch starts)
    __shared__ int s_wsum[BT / 64];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int e = blockIdx.x * BT + tid;
    if (blockIdx.x == 0 && tid < 3 && next_queue) next_queue[tid] = 0u;
    int k = 0, r = 0;
    const uint64_t* lsrc = nullptr;        // where the post-step legal words come from
    int clear_bit = -1;
    bool was_reset = false;
    if (e < b.n_envs) {
        const int nl = b.nlegal[e];
        r = b.env_region[e];
        was_reset = (nl == 0);
        if (nl == 0) {
            if (b.auto_reset) {            // xr_env_reset(rotate = 1), examples/launch_training.py:37-46
                if (b.env_replay[e] == b.max_route_count) r = (int)(((int64_t)r + b.n_envs) % b.n_regions);
                const XrRegionDev R = b.regions[r];
                lsrc = b.legal0 + R.legal0_off;
                k = R.nlegal0;
            }
        } else {
            const XrRegionDev R = b.regions[r];
            const int a = actions[e];
            lsrc = b.legal + (int64_t)e * b.legal_words;
            k = nl;
            if (a >= 1 && a <= R.n_nets && ((lsrc[(a - 1) >> 6] >> ((a - 1) & 63)) & 1ULL)) {
                clear_bit = a - 1;
                k = nl - 1;
            }
        }
        b.plan_region[e] = r;
    }
    // the step kernel keeps the lowest XR_SPLIT_KEEP ranks of every env, the writer takes the rest.
    // In-place form (xr_batch_step_observe_inplace: the caller's buffer still holds the previous observation): the planes of a
    // net are static, and removing net a from the ascending net list shifts only the nets ABOVE it down by one slot — the
    // nets below keep their slot and their bytes.  So only the remaining nets above the routed one are units; a slot that
    // re-initialises writes all of them, a rejected action none.
    int kskip = XR_SPLIT_KEEP(b, k);
    if (b.obs_incremental && e < b.n_envs && !was_reset) {
        if (clear_bit < 0) kskip = k;
        else {
            int below = 0;
            for (int w = 0; w <= (clear_bit >> 6); w++) {
                uint64_t m = lsrc[w];
                if (w == (clear_bit >> 6)) m &= (1ULL << (clear_bit & 63)) - 1ULL;
                below += __popcll(m);
            }
            kskip = below;
        }
    }
    const int kw = k - kskip;
    int incl = kw;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) s_wsum[wv] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wv; w++) woff += s_wsum[w];
    if (tid == BT - 1) s_base = (int)atomicAdd(&b.queue[2], (unsigned)(woff + incl));
    __syncthreads();
    const int off = s_base + woff + incl - kw;
    if (e < b.n_envs) {
        int j = 0;
        for (int w = 0; w < b.legal_words && j < k; w++) {
            uint64_t m = lsrc[w];
            if (clear_bit >= 0 && (clear_bit >> 6) == w) m &= ~(1ULL << (clear_bit & 63));
            while (m) {
                const int bit = __ffsll((unsigned long long)m) - 1;
                m &= m - 1;
                if (j >= kskip) {
                    b.plan_units[off + j - kskip] = ((uint32_t)e << 14) | (uint32_t)j;
                    b.plan_unit_net[off + j - kskip] = (w << 6) + bit + 1;
                }
                j++;
            }
        }
    }
    if (order) {                // (gridDim.x == 1, n_envs <= BT) counting sort of the env slots by descending work class: xr_route_order_kernel
        __shared__ uint32_t s_cnt[256];
        __shared__ uint32_t s_csum[4];
        if (tid < 256) s_cnt[tid] = 0;
        __syncthreads();
        const int cls = e < b.n_envs ? xr_route_work_class(b, actions, e) : -1;
        if (cls >= 0) atomicAdd(&s_cnt[cls], 1u);
        __syncthreads();
        uint32_t v = 0, inc = 0;
        if (tid < 256) {
            v = s_cnt[255 - tid];
            inc = v;
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t u = __shfl_up(inc, o, 64);
                if (lane >= o) inc += u;
            }
            if (lane == 63) s_csum[wv] = inc;
        }
        __syncthreads();
        if (tid < 256) {
            uint32_t base = 0;
            for (int w = 0; w < wv; w++) base += s_csum[w];
            s_cnt[255 - tid] = base + inc - v;
        }
        __syncthreads();
        if (cls >= 0) order[atomicAdd(&s_cnt[cls], 1u)] = e;
    }
}

#ifdef XR_NP_PLAIN
#define XR_NP_ST4(ptr, val) (*reinterpret_cast<float4*>(ptr) = (val))
#else
#define XR_NP_ST4(ptr, val) XR_ST4(ptr, val)
#endif
#define XR_NP_J 9            // float4 groups per thread per tile: 256 threads * 4 nodes * 9 = 9216 nodes
// one unit = the 7 planes of one remaining net of one env (aligned planes: every region's N % 4 == 0)
__device__ __forceinline__ void xr_unit_aligned(const XrBatchDev& b, int u) {
    const int tid = threadIdx.x;
    const uint32_t ent = b.plan_units[u];
    const int id = b.plan_unit_net[u];
    const int e = (int)(ent >> 14), rank = (int)(ent & 0x3FFFu);
    const XrRegionDev R = b.regions[b.plan_region[e]];
    const int N = R.N, Z = R.Z, Y = R.Y, X = R.X, YZ = Y * Z;
    const int16_t* __restrict__ nn = b.rg_node_net + R.node_off;
    float* __restrict__ out = b.obs_out + (int64_t)e * b.obs_stride + (int64_t)(2 + 7 * rank) * N;
    const int ngrp = N >> 2;                                   // N % 4 == 0 in this mode
    // The 7 planes of a net are zero except at its access points (a handful of nodes): the masks come from the net's
    // access-point list (ap_feat: node | has-same-net-axis-neighbour << 31, computed at load), not from a scan of the region's
    // node array — a unit then reads ~50 bytes instead of 17 KB (round 1: 0.6 GB of HBM reads per 4096-env launch).
    const int ap_lo = b.net_csr[R.net_off + id], ap_hi = b.net_csr[R.net_off + id + 1];
    (void)nn; (void)X; (void)Y; (void)Z; (void)YZ;
    for (int g0 = 0; g0 < ngrp; g0 += 256 * XR_NP_J) {
        unsigned bits[XR_NP_J];                                // per group: AP mask (bits 0..3), neighbour mask (4..7)
#pragma unroll
        for (int j = 0; j < XR_NP_J; j++) bits[j] = 0;
        for (int i = ap_lo; i < ap_hi; i++) {
            const int v = b.ap_feat[R.ap_off + i];             // wave-uniform address: one scalar load
            const int f = v & 0x7FFFFFFF;
            const unsigned m = (1u | (v < 0 ? 16u : 0u)) << (f & 3);
            const int g = (f >> 2) - g0 - tid;                 // == j * 256 for the thread and slot that hold node f
#pragma unroll
            for (int j = 0; j < XR_NP_J; j++) bits[j] |= (g == j * 256) ? m : 0u;
        }
        // plane 0 of the net: AP mask; planes 1..6: the six aliased "has a same-net axis neighbour" planes
#ifndef XR_NP_ORDER
#define XR_NP_ORDER 1
#endif
#if XR_NP_ORDER == 0       // plane-major: the workgroup writes the unit as one sequential run
#pragma unroll 1
        for (int pl = 0; pl < 7; pl++) {
            float* __restrict__ pp = out + (int64_t)pl * N;
            const int sh = pl ? 4 : 0;
#pragma unroll
            for (int j = 0; j < XR_NP_J; j++) {
                const int g = g0 + j * 256 + tid;
                if (g < ngrp) {
                    const unsigned m = bits[j] >> sh;
                    float4 v;
                    v.x = (m & 1u) ? 1.f : 0.f; v.y = (m & 2u) ? 1.f : 0.f;
                    v.z = (m & 4u) ? 1.f : 0.f; v.w = (m & 8u) ? 1.f : 0.f;
                    XR_NP_ST4(pp + ((int64_t)g << 2), v);
                }
            }
        }
#else                      // rotation: 4 KB of each of the 7 planes in turn (the step kernel's own order)
#pragma unroll
        for (int j = 0; j < XR_NP_J; j++) {
            const int g = g0 + j * 256 + tid;
            if (g < ngrp) {
                float* __restrict__ pp = out + ((int64_t)g << 2);
                const unsigned m0 = bits[j], m1 = bits[j] >> 4;
                float4 v0, v1;
                v0.x = (m0 & 1u) ? 1.f : 0.f; v0.y = (m0 & 2u) ? 1.f : 0.f; v0.z = (m0 & 4u) ? 1.f : 0.f; v0.w = (m0 & 8u) ? 1.f : 0.f;
                v1.x = (m1 & 1u) ? 1.f : 0.f; v1.y = (m1 & 2u) ? 1.f : 0.f; v1.z = (m1 & 4u) ? 1.f : 0.f; v1.w = (m1 & 8u) ? 1.f : 0.f;
                XR_NP_ST4(pp, v0);
#pragma unroll
                for (int pl = 1; pl < 7; pl++) XR_NP_ST4(pp + (int64_t)pl * N, v1);
            }
        }
#endif
    }
}

__global__ void __launch_bounds__(256) xr_netplane_kernel(XrBatchDev b) {
    const int total = (int)b.queue[2];
    for (int u = blockIdx.x; u < total; u += gridDim.x) xr_unit_aligned(b, u);
}

// The same units for regions whose N is not a multiple of 4 (design-derived regions): a unit's 7*N floats start at
// float (2+7*rank)*N of the env's row, 16-byte aligned only by chance.  Per unit the workgroup reduces the net's two
// node masks to one byte per node in LDS, then writes the unit's float range as aligned float4 slots (plane and node
// of a slot resolved per float where a slot straddles two planes) plus at most 3 scalar floats at either end.
// s_m: LDS, (n_max / 16 + 2) words.  Two bits per node, 16 nodes per word: bit 2i = node is an AP of the net,
// bit 2i+1 = ... with a same-net neighbour.  Ends with a barrier (s_m may be reused at once).
__device__ __forceinline__ void xr_unit_stream(const XrBatchDev& b, int u, uint32_t* s_m) {
    const int tid = threadIdx.x;
    const uint32_t ent = b.plan_units[u];
    const int id = b.plan_unit_net[u];
    const int e = (int)(ent >> 14), rank = (int)(ent & 0x3FFFu);
    const XrRegionDev R = b.regions[b.plan_region[e]];
    const int N = R.N, Z = R.Z, Y = R.Y, X = R.X, YZ = Y * Z;
    const int nwords = (N + 15) >> 4;
    (void)X; (void)Y; (void)Z; (void)YZ;
    // The two masks are zero except at the net's access points (a handful of nodes): they come from the net's access-point list
    // (ap_feat: node | has-same-net-axis-neighbour << 31, decided once at load), exactly as xr_unit_aligned takes them — not
    // from a scan of the region's node array (17 KB and 16 compares per word, per unit: what kept this writer at ~0.5 of peak)
    const int ap_lo = b.net_csr[R.net_off + id], ap_hi = b.net_csr[R.net_off + id + 1];
    int my_ap = 0;
    const bool has_ap = ap_lo + tid < ap_hi;                   // (<= XR_MAX_AP_PER_NET = 128 access points: one per thread)
    if (has_ap) my_ap = b.ap_feat[R.ap_off + ap_lo + tid];      // issued before the zero fill
    for (int w = tid; w <= nwords; w += 256) s_m[w] = 0u;      // one spare word: the funnel shift reads w + 1
    xr_lds_barrier();
    if (has_ap) {
        const int f = my_ap & 0x7FFFFFFF;
        atomicOr(&s_m[f >> 4], (my_ap < 0 ? 3u : 1u) << ((f & 15) << 1));
    }
    xr_lds_barrier();
    // The unit's 7 planes are ONE contiguous run of 7*N floats at float (2 + 7*rank)*N of the env's row — plane boundaries only
    // matter for the VALUE of a float (plane = offset / N, node = offset % N), not for where it is written.  So the run is written
    // exactly like an aligned unit: cut at 128-byte-aligned multiples of 4 KB into 7 pieces, slot t of every piece in rotation,
    // every wave store 8 whole 128-byte lines.  (tools/micro/write_unaligned.hip, N = 7650: 16-byte-aligned slots that start
    // anywhere in a line reach 4.3-4.6 TB/s, the same bytes as whole lines 5.6-5.7 TB/s — as fast as aligned planes.)
    float* __restrict__ run = b.obs_out + (int64_t)e * b.obs_stride + (int64_t)(2 + 7 * rank) * N;
    const int total = 7 * N;
    const int head = (int)((32u - (uint32_t)((reinterpret_cast<uintptr_t>(run) >> 2) & 31u)) & 31u);     // floats before the first 128-byte line
    const int nslot = total > head ? (total - head) >> 2 : 0;                 // aligned float4 slots of the run
    const int piece = ((nslot + 7 * 256 - 1) / (7 * 256)) * 256;              // slots per piece: a multiple of 256 (4 KB)
    auto one = [&](int o) -> float {          // value of the float at run offset o
        int pl = 0, f = o;
        while (f >= N) { f -= N; pl++; }
        return ((s_m[f >> 4] >> (((f & 15) << 1) + (pl ? 1 : 0))) & 1u) ? 1.f : 0.f;
    };
    // (plane, node) of this thread's slot in each piece, advanced by 1024 floats per step
    int ppl[7], pnd[7];
#pragma unroll
    for (int p = 0; p < 7; p++) {
        int o = head + ((p * piece + tid) << 2), pl = 0;
        while (o >= N && pl < 7) { o -= N; pl++; }
        ppl[p] = pl; pnd[p] = o;
    }
    for (int it = 0; it < piece; it += 256) {
#pragma unroll
        for (int p = 0; p < 7; p++) {
            const int sl = p * piece + it + tid;
            if (sl < nslot) {
                const int f = pnd[p], pl = ppl[p];
                float4 v;
                if (f + 3 < N) {                                  // four nodes of one plane: two mask words, one funnel shift
                    const int bit = f << 1, w = bit >> 5;
                    const uint32_t m = __funnelshift_r(s_m[w], s_m[w + 1], bit & 31) >> (pl ? 1 : 0);
                    v.x = (m & 1u) ? 1.f : 0.f; v.y = (m & 4u) ? 1.f : 0.f; v.z = (m & 16u) ? 1.f : 0.f; v.w = (m & 64u) ? 1.f : 0.f;
                } else {                                          // the slot straddles two planes: float by float
                    const int o = head + (sl << 2);
                    v.x = one(o); v.y = one(o + 1); v.z = one(o + 2); v.w = one(o + 3);
                }
                XR_ST4(run + head + (sl << 2), v);
            }
            int nf = pnd[p] + 1024;
            while (nf >= N) { nf -= N; ppl[p]++; }                // (N < 1024: several planes per step)
            pnd[p] = nf;
        }
    }
    // ragged ends of the run: < 32 floats before the first whole line, < 4 after the last slot
    if (tid < head && tid < total) run[tid] = one(tid);
    const int tail0 = head + (nslot << 2);
    if (tid >= 64 && tail0 + (tid - 64) < total) run[tail0 + (tid - 64)] = one(tail0 + (tid - 64));
    xr_lds_barrier();
}

#ifndef XR_QUEUE_SKIP
#define XR_QUEUE_SKIP 1
#endif
#ifndef XR_QUEUE_BATCH
#define XR_QUEUE_BATCH 1
#endif
// Static first tasks of the queue form (see xr_step_queue_kernel): of G workgroups, those with bit `sh` of their index clear start with
// a route (first_r of them), the others with a unit (first_u); sh < 0: every workgroup starts with a route.
__device__ __forceinline__ void xr_queue_first(int G, int sh, int& first_r, int& first_u) {
    if (sh < 0) { first_r = G; first_u = 0; return; }
    first_r = ((G >> (sh + 1)) << sh) + min(G & ((2 << sh) - 1), 1 << sh);
    first_u = G - first_r;
}

// Helper writers of the queue form: workgroups WITHOUT LDS that drain the same unit queue (queue[1]) as the persistent step
// kernel, launched concurrently on an internal stream.  The step kernel's workgroups are capped at 4 per CU by the router's LDS
// and spend ~45 % of their time routing, so on average only ~2.2 of them per CU are writing — not enough stores in flight for the
// HBM write ceiling of this pattern (1024 pure writers reach it).  The helpers fill the wave slots the LDS cap leaves free.
__global__ void __launch_bounds__(256) xr_unit_helper_kernel(XrBatchDev b) {
    __shared__ int s_u;
    const int total = (int)b.queue[2];
    int first_r, first_u;
    xr_queue_first(b.queue_grid, XR_QUEUE_SKIP == 1 ? b.queue_skip_shift : -1, first_r, first_u);     // (units 0 .. first_u - 1 belong to the step kernel)
    for (;;) {
        if (threadIdx.x == 0) s_u = first_u + (int)atomicAdd(&b.queue[1], 1u);
        __syncthreads();
        const int u = s_u;
        __syncthreads();
        if (u >= total) break;
        xr_unit_aligned(b, u);
    }
}

__global__ void __launch_bounds__(256) xr_netplane_stream_kernel(XrBatchDev b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int total = (int)b.queue[2];
    for (int u = blockIdx.x; u < total; u += gridDim.x) xr_unit_stream(b, u, reinterpret_cast<uint32_t*>(smem));
}

// ------------------------------------------------------------------------------------------------
// queue form of the step (XR_OBS_QUEUE): ONE persistent launch (as many workgroups as fit the chip) that drains two task
// queues — envs to route (+ their planes 0..1) and net-plane units to write.  Units do not depend on routing
// (xr_plan_kernel), so they are the filler that keeps HBM busy while other workgroups route, and the tail of the
// launch is one 20 us unit instead of one whole env.  A workgroup alternates one route task with `quota` units (a little
// under the average units per env, so that the routes run out first and the launch ends in a pure-write drain);
// half of the workgroups (bit 5 of the workgroup index: spread over every XCD and CU) start with units so that the launch writes
// from its first microseconds.  No workgroup ever waits for another one.
// ------------------------------------------------------------------------------------------------
#ifndef XR_QUEUE_WAVES_PER_SIMD
#define XR_QUEUE_WAVES_PER_SIMD 4      // register budget of the persistent step kernel: 4 waves per SIMD = its own 4 workgroups per CU.
#endif                                 // (6 = <= 80 VGPRs, room for helper-writer waves beside them: measured no faster)
template <bool LDS_DIST, int ZCH>
__global__ void __launch_bounds__(1024, XR_QUEUE_WAVES_PER_SIMD) xr_step_queue_kernel(XrBatchDev b, const int32_t* __restrict__ actions) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_task;
    const int tid = threadIdx.x;
    const int B = b.n_envs;
    const int total = (int)b.queue[2];
    const int quota = max(1, (int)(((int64_t)total * b.queue_quota_pm) / (1000 * (int64_t)B)));
    bool routes_left = true, units_left = total > 0;
    bool skip_route = XR_QUEUE_SKIP == 1 ? (b.queue_skip_shift >= 0 && ((blockIdx.x >> b.queue_skip_shift) & 1) != 0) : XR_QUEUE_SKIP == 2 ? (blockIdx.x & 3) == 3
                    : XR_QUEUE_SKIP == 3 ? (blockIdx.x & 3) != 0 : false;
    // The FIRST task of every workgroup is static — the route-first workgroups take route tasks 0, 1, 2, ... by their rank among
    // themselves, the unit-first ones units 0, 1, 2, ... — and the two counters hand out what follows: a thousand workgroups opening
    // the launch with an atomic on the same word serialise in L2 (tools/queue_timeline_probe.py: the first routes started 6-12 us into
    // the launch).
    int first_r = 0, first_u = 0, my_first = -1;          // tasks handed out statically; this workgroup's own one (-1: taken)
    const bool unit_first = skip_route;
    if (XR_QUEUE_SKIP == 1) {
        const int sh = b.queue_skip_shift, G = (int)gridDim.x, i = (int)blockIdx.x;
        xr_queue_first(G, sh, first_r, first_u);
        my_first = sh < 0 ? i : ((i >> (sh + 1)) << sh) + (i & ((1 << sh) - 1));          // rank among the workgroups of its kind
    }
#ifdef XR_TIMELINE     // per workgroup (slot blockIdx.x of phase_cycles): start, end, time routing, time writing units, routes, units,
                       // time of the last route's end (100 MHz ticks)
    long long tl_start = wall_clock64(), tl_route = 0, tl_unit = 0, tl_last_route = 0;
    int tl_nr = 0, tl_nu = 0;
#define XR_TL_NOW() wall_clock64()
#endif
    while (routes_left || units_left) {
        if (routes_left && !skip_route) {
            if (tid == 0) {
                const int t = (my_first >= 0 && !unit_first) ? my_first : first_r + (int)atomicAdd(&b.queue[0], 1u);
                s_task = (t < B && b.route_order) ? b.route_order[t] : t;      // (small batches: longest predicted route first)
            }
            if (!unit_first) my_first = -1;
            xr_lds_barrier();
            const int e = s_task;
            xr_lds_barrier();
            if (e < B) {
#ifdef XR_TIMELINE
                const long long t0 = XR_TL_NOW();
#endif
#ifdef XR_ROUTE_PRIO      // A/B: routing waves ahead of the unit writers they share a CU with (instruction arbitration)
                __builtin_amdgcn_s_setprio(XR_ROUTE_PRIO);
#endif
                xr_route_dispatch<LDS_DIST, ZCH>(b, e, actions[e], smem);
#ifdef XR_ROUTE_PRIO
                __builtin_amdgcn_s_setprio(0);
#endif
                xr_obs_epilogue(b, e, smem, true);
                xr_lds_barrier();
#ifdef XR_TIMELINE
                tl_last_route = XR_TL_NOW(); tl_route += tl_last_route - t0; tl_nr++;
#endif
            } else {
                routes_left = false;
            }
        }
        skip_route = false;
        if (units_left) {
            // the next claim is issued before the current unit is written (an atomic round trip costs ~4 us, a unit ~20 us);
            // claiming several units at once measured slower (XR_QUEUE_BATCH 2: +1.5 %, 4: +7 %)
            const int nb = routes_left ? max(1, quota / XR_QUEUE_BATCH) : (1 << 30);
            unsigned nx = 0;
            const unsigned ubase = (unsigned)first_u * (unsigned)XR_QUEUE_BATCH;
            if (tid == 0) nx = (my_first >= 0 && unit_first) ? (unsigned)my_first * (unsigned)XR_QUEUE_BATCH
                                                             : ubase + atomicAdd(&b.queue[1], (unsigned)XR_QUEUE_BATCH);
            if (unit_first) my_first = -1;
            for (int i = 0; i < nb; i++) {
                if (tid == 0) s_task = (int)nx;
                xr_lds_barrier();
                const int u0 = s_task;
                xr_lds_barrier();
                if (u0 >= total) { units_left = false; break; }
                if (tid == 0 && i + 1 < nb) nx = ubase + atomicAdd(&b.queue[1], (unsigned)XR_QUEUE_BATCH);     // used after this batch
#ifdef XR_TIMELINE
                const long long t0 = XR_TL_NOW();
#endif
                const int u1 = min(u0 + XR_QUEUE_BATCH, total);
                for (int u = u0; u < u1; u++) {
                    if (b.obs_vec4 == 1) xr_unit_aligned(b, u);
                    else xr_unit_stream(b, u, reinterpret_cast<uint32_t*>(smem));
                }
#ifdef XR_TIMELINE
                tl_unit += XR_TL_NOW() - t0; tl_nu += u1 - u0;
#endif
            }
            xr_lds_barrier();
        }
    }
#ifdef XR_TIMELINE
    if (tid == 0 && (int)blockIdx.x < B) {
        long long* o = b.phase_cycles + (int64_t)blockIdx.x * 8;
        o[0] = tl_start; o[1] = XR_TL_NOW(); o[2] = tl_route; o[3] = tl_unit; o[4] = tl_nr; o[5] = tl_nu; o[6] = tl_last_route; o[7] = 1;
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// whole-order re-route: the A3C / MCTS simulator contract.  The reference's A3C env answers with a complete
// net list (baseline/A3C/utils.py:305-307 `message.response.net_list.extend(action_list)`) and its MCTS
// dispatcher re-routes the region from scratch with `routed_nets + unrouted_nets` after every selection
// (baseline/xroute/trainer4/dispatcher.py:113-118).  One workgroup restores its env to the region's initial
// state and routes the listed nets back to back; per-net metric deltas (the simulator's `metrics_delta`,
// net_ordering.proto v2 field 13) and a per-net route counter (`count_map`, field 12) go to net_stats.
// ------------------------------------------------------------------------------------------------
template <bool LDS_DIST, int ZCH>
__global__ void xr_order_kernel(XrBatchDev b, const int32_t* __restrict__ orders, int stride,
                                int32_t* __restrict__ net_stats) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int e = blockIdx.x;
    const int tid = threadIdx.x;
    xr_env_reset(b, e, 0, 0);
    __syncthreads();
    int st_acc = 0, plen_acc = 0, sweeps_acc = 0;
    const int32_t* __restrict__ ord = orders + (int64_t)e * stride;
    for (int k = 0; k < stride; k++) {
        const int a = ord[k];
        if (a <= 0) break;                         // list terminator (uniform)
        if (b.nlegal[e] == 0) break;               // everything routed: the rest of the list is ignored
        xr_route_dispatch<LDS_DIST, ZCH>(b, e, a, smem);
        __syncthreads();
        if (tid == 0) {
            const int st = b.status[e];
            st_acc |= st;
            if (!(st & XR_ENV_BAD_ACTION)) {
                plen_acc += b.path_len[e];
                sweeps_acc += b.sweeps[e];
                if (net_stats) {
                    int32_t* s = net_stats + ((int64_t)e * stride + (a - 1)) * 4;
                    s[0] = b.delta[3 * e + 0]; s[1] = b.delta[3 * e + 1]; s[2] = b.delta[3 * e + 2];
                    s[3] += 1;
                }
            }
        }
    }
    if (tid == 0) {
        const XrRegionDev R = b.regions[b.env_region[e]];
        const int dv = b.cum[3 * e + 0] - R.m0[0], dw = b.cum[3 * e + 1] - R.m0[1], dvia = b.cum[3 * e + 2] - R.m0[2];
        b.delta[3 * e + 0] = dv; b.delta[3 * e + 1] = dw; b.delta[3 * e + 2] = dvia;
        b.reward[e] = -1.0 * (b.w_violation * (double)dv + b.w_via * (double)dvia + b.w_wirelength * (double)dw);
        b.status[e] = st_acc;
        b.path_len[e] = plen_acc;
        b.sweeps[e] = sweeps_acc;
        xr_publish_record(b, e);
    }
}

// ------------------------------------------------------------------------------------------------
// compact-consumer mode: the 7 planes of a (region, net) pair on demand.  They are functions of the region's static node
// array only (build_3Dgrid.py:106-142: access-point mask + the aliased "has a same-net axis neighbour" plane), so a consumer
// that caches per-(region, net) results (agents.NetVectorCache; the reference consumer baseline/DQN/DQN.py:138-155 re-encodes
// every net at every step) needs them ONCE per episode set, while the step writes planes 0..1 only (xr_batch_step_compact).
// One workgroup per pair; float4 stores when the planes are 16-byte aligned.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) xr_netplanes_pairs_kernel(XrBatchDev b, const int32_t* __restrict__ pair_region,
                                                               const int32_t* __restrict__ pair_net, float* __restrict__ out,
                                                               int64_t pair_stride, int vec4) {
    const int p = blockIdx.x;
    const int r = pair_region[p], id = pair_net[p];
    float* __restrict__ o = out + (int64_t)p * pair_stride;
    if (r < 0 || r >= b.n_regions) return;
    const XrRegionDev R = b.regions[r];
    const int N = R.N, Z = R.Z, Y = R.Y, X = R.X, YZ = Y * Z;
    const int16_t* __restrict__ nn = b.rg_node_net + R.node_off;
    auto feat = [&](int f) -> unsigned {          // bit 0: access point of the net, bit 1: ... with a same-net axis neighbour
        if (f >= N || id <= 0 || nn[f] != id) return 0u;
        const int z = f % Z, y = (f / Z) % Y, x = f / YZ;
        const bool adj = (x + 1 < X && nn[f + YZ] == id) || (y > 0 && nn[f - Z] == id) || (x > 0 && nn[f - YZ] == id) ||
                         (y + 1 < Y && nn[f + Z] == id) || (z + 1 < Z && nn[f + 1] == id) || (z > 0 && nn[f - 1] == id);
        return adj ? 3u : 1u;
    };
    if (vec4 && (N & 3) == 0) {
        for (int f0 = threadIdx.x * 4; f0 < N; f0 += 256 * 4) {
            unsigned m[4];
#pragma unroll
            for (int j = 0; j < 4; j++) m[j] = feat(f0 + j);
            const float4 v0 = make_float4((m[0] & 1u) ? 1.f : 0.f, (m[1] & 1u) ? 1.f : 0.f, (m[2] & 1u) ? 1.f : 0.f, (m[3] & 1u) ? 1.f : 0.f);
            const float4 v1 = make_float4((m[0] & 2u) ? 1.f : 0.f, (m[1] & 2u) ? 1.f : 0.f, (m[2] & 2u) ? 1.f : 0.f, (m[3] & 2u) ? 1.f : 0.f);
            XR_ST4(o + f0, v0);
#pragma unroll
            for (int pl = 1; pl < 7; pl++) XR_ST4(o + (int64_t)pl * N + f0, v1);
        }
    } else {
        for (int f = threadIdx.x; f < N; f += 256) {
            const unsigned m = feat(f);
            o[f] = (m & 1u) ? 1.f : 0.f;
            const float a = (m & 2u) ? 1.f : 0.f;
#pragma unroll
            for (int pl = 1; pl < 7; pl++) o[(int64_t)pl * N + f] = a;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// XR-Maze v2: the guide of every (region, net) as a static bitmask (round 5).  Whether a node lies inside a net's guide — the boxes
// xr_batch_load_guides handed over, inflated by guide_margin, or the bounding box of the net's access points when it has none
// (xr_guide_load) — never changes while the regions are loaded, yet round 4's router decided it per ROUTE: two dependent
// global loads for the boxes and a pass over the nodes with up to 8 box tests each (~13 % of a route on the design-derived pack).  One
// workgroup per (region, net), once per guide load: bit j of byte c = node 8 c + j is OUTSIDE the guide.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) xr_guide_mask_kernel(XrBatchDev b, uint8_t* __restrict__ masks, int region_base) {
    __shared__ int s_bb[4], s_ngb;
    __shared__ int4 s_gbx[XR_GUIDE_MAX_BOXES];
    const int r = region_base + blockIdx.y, a = blockIdx.x + 1, tid = threadIdx.x;
    const XrRegionDev R = b.regions[r];
    if (a > R.n_nets) return;
    const int Y = R.Y, Z = R.Z, YZ = Y * Z, N = R.N;
    if (tid == 0) { s_bb[0] = 0x7FFFFFFF; s_bb[1] = -1; s_bb[2] = 0x7FFFFFFF; s_bb[3] = -1; }
    __syncthreads();
    const int lo = b.net_csr[R.net_off + a], hi = b.net_csr[R.net_off + a + 1];
    for (int i = lo + tid; i < hi; i += 256) {
        const int apf = b.ap_node[R.ap_off + i];
        const int gy = (apf / Z) % Y, gx = apf / YZ;
        atomicMin(&s_bb[0], gx); atomicMax(&s_bb[1], gx); atomicMin(&s_bb[2], gy); atomicMax(&s_bb[3], gy);
    }
    __syncthreads();
    xr_guide_load(b, R, a, s_bb, Z, s_gbx, &s_ngb, tid);
    __syncthreads();
    const int ngb = s_ngb;
    const int4 g0 = s_gbx[0];
    uint8_t* __restrict__ row = masks + R.gmask_off + (int64_t)(a - 1) * R.gmask_stride;
    for (int c = tid; c < R.gmask_stride; c += 256) {
        uint32_t out = 0;
        for (int j = 0; j < 8; j++) {
            const int f = (c << 3) + j;
            if (f < N) {
                const int z = f % Z, y = (f / Z) % Y, x = f / YZ;
                out |= (uint32_t)(!xr_guide_has(s_gbx, ngb, g0, x, y, z)) << j;
            }
        }
        row[c] = (uint8_t)out;
    }
}

// ------------------------------------------------------------------------------------------------
// Compact state for a central learner (BASELINE config 4: "gather compact state and expand on the learner GPU").
// The two planes of an observation that change per step are functions of very little: plane 0 ("blockage or used",
// build_3Dgrid.py:19-36,94-103) is one BIT per node, plane 1 (the remaining nets' ids ascending at flat positions 0..K-1,
// :144-161) IS the legal-net bitmask.  One row per env slot, fixed size, ready for ONE all_gather:
//     int32 region (+ region_base: the sender's local region index as an index into the learner's region table)
//     int32 nlegal | int32 legal_words | int32 occ_words
//     uint64 legal[legal_words] | uint64 occ[occ_words]      bit f % 64 of word f / 64 = node f (flat observation order)
// = 16 + 8·(legal_words + ceil(n_max / 64)) bytes: 1.1 KB for an ispd18_test1-sized region against 69 KB for the two fp32 planes.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) xr_pack_state_kernel(XrBatchDev b, uint8_t* __restrict__ rows, int64_t row_bytes, int region_base) {
    const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int r = b.env_region[e];
    const XrRegionDev R = b.regions[r];
    uint8_t* row = rows + (int64_t)e * row_bytes;
    uint64_t* lg = reinterpret_cast<uint64_t*>(row + 16);
    uint64_t* occ = lg + b.legal_words;
    const int ow = (b.n_max + 63) >> 6;
    if (tid == 0) {
        int32_t* hdr = reinterpret_cast<int32_t*>(row);
        hdr[0] = r + region_base; hdr[1] = b.nlegal[e]; hdr[2] = b.legal_words; hdr[3] = ow;
    }
    for (int w = tid; w < b.legal_words; w += 256) lg[w] = b.legal[(int64_t)e * b.legal_words + w];
    // eight nodes per lane (two 16-byte loads), eight lanes per 64-bit word: a wave covers 512 nodes per pass
    const int16_t* __restrict__ nn = b.rg_node_net + R.node_off;
    const int16_t* __restrict__ own = b.owner + (int64_t)e * b.n_max;
    const int nchunk = ow << 3;                       // chunks of 8 nodes in the row (the state rows are padded to multiples of 8)
    const int cmax = (R.N + 7) >> 3;
    for (int c = tid; c < ((nchunk + 255) & ~255); c += 256) {
        uint32_t bits = 0;
        if (c < cmax) {
            const int4 vn = *reinterpret_cast<const int4*>(nn + (c << 3));
            const int4 vo = *reinterpret_cast<const int4*>(own + (c << 3));
            const uint32_t n4[4] = {(uint32_t)vn.x, (uint32_t)vn.y, (uint32_t)vn.z, (uint32_t)vn.w};
            const uint32_t o4[4] = {(uint32_t)vo.x, (uint32_t)vo.y, (uint32_t)vo.z, (uint32_t)vo.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                bits |= (uint32_t)(((n4[j] & 0xFFFFu) == 0xFFFFu) || (o4[j] & 0xFFFFu)) << (2 * j);
                bits |= (uint32_t)(((n4[j] >> 16) == 0xFFFFu) || (o4[j] >> 16)) << (2 * j + 1);
            }
            const int left = R.N - (c << 3);
            if (left < 8) bits &= (1u << left) - 1u;
        }
        uint64_t v = (uint64_t)bits << (8 * (lane & 7));
        v |= __shfl_xor(v, 1, 64); v |= __shfl_xor(v, 2, 64); v |= __shfl_xor(v, 4, 64);
        if ((lane & 7) == 0 && (c >> 3) < ow) occ[c >> 3] = v;
    }
}

// rows -> the fp32 head rows xr_batch_step_compact writes (planes 0..1, each env in its own region's layout), on the learner's device; `b` only
// supplies the region table (the learner's batch holds every region of the job).  A row that does not parse (region index outside the table,
// sizes that do not fit, a nets-left count that is not the popcount of its mask) is left unwritten and flagged nlegal = region = -1.
__global__ void __launch_bounds__(256) xr_expand_state_kernel(XrBatchDev b, const uint8_t* __restrict__ rows, int64_t row_bytes, float* __restrict__ head,
                                                            int64_t head_stride, int32_t* __restrict__ nlegal_out, int32_t* __restrict__ region_out, int vec4) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* s_ids = reinterpret_cast<int*>(smem);
    int* s_pref = s_ids + b.legal_words * 64;
    const int i = blockIdx.x, tid = threadIdx.x;
    const uint8_t* row = rows + (int64_t)i * row_bytes;
    const int32_t* hdr = reinterpret_cast<const int32_t*>(row);
    const int r = hdr[0], nl = hdr[1], lw = hdr[2], ow = hdr[3];
    bool ok = r >= 0 && r < b.n_regions && lw >= 0 && lw <= b.legal_words && ow >= 0 && 16 + 8 * ((int64_t)lw + ow) <= row_bytes;
    XrRegionDev R = b.regions[ok ? r : 0];
    ok = ok && (int64_t)ow * 64 >= R.N && 2 * (int64_t)R.N <= head_stride;
    const uint64_t* lg = reinterpret_cast<const uint64_t*>(row + 16);
    const uint64_t* occ = lg + (ok ? lw : 0);
    int K = 0;
    if (ok) {                  // (uniform over the workgroup: every thread read the same header)
        K = xr_legal_ids(lg, lw, s_ids, s_pref);
        ok = K == nl && K <= R.n_nets && (K == 0 || s_ids[K - 1] <= R.n_nets);
    }
    if (tid == 0) { nlegal_out[i] = ok ? K : -1; region_out[i] = ok ? r : -1; }
    if (!ok) return;
    const int N = R.N;
    float* __restrict__ o = head + (int64_t)i * head_stride;
    if (vec4 && (N & 3) == 0) {
        for (int f0 = tid * 4; f0 < N; f0 += 1024) {
            const uint32_t m = (uint32_t)(occ[f0 >> 6] >> (f0 & 63)) & 15u;
            XR_ST4(o + f0, make_float4((m & 1u) ? 1.f : 0.f, (m & 2u) ? 1.f : 0.f, (m & 4u) ? 1.f : 0.f, (m & 8u) ? 1.f : 0.f));
            XR_ST4(o + N + f0, make_float4(f0 < K ? (float)s_ids[f0] : 0.f, f0 + 1 < K ? (float)s_ids[f0 + 1] : 0.f,
                                           f0 + 2 < K ? (float)s_ids[f0 + 2] : 0.f, f0 + 3 < K ? (float)s_ids[f0 + 3] : 0.f));
        }
    } else {
        for (int f = tid; f < N; f += 256) {
            o[f] = ((occ[f >> 6] >> (f & 63)) & 1ull) ? 1.f : 0.f;
            o[N + f] = f < K ? (float)s_ids[f] : 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host-callable launchers (kept here so that only this TU needs the <<<>>> syntax)
// ------------------------------------------------------------------------------------------------
extern "C" {

hipError_t xr_launch_ingest(const uint32_t* rec, int16_t* node_net, int16_t* owner0, int64_t total, hipStream_t st) {
    if (total <= 0) return hipSuccess;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(xr_ingest_kernel, dim3(blocks), dim3(256), 0, st, rec, node_net, owner0, total);
    return hipGetLastError();
}

hipError_t xr_launch_reset(const XrBatchDev* b, const uint8_t* mask, int rotate, hipStream_t st) {
    hipLaunchKernelGGL(xr_reset_kernel, dim3(b->n_envs), dim3(256), 0, st, *b, mask, rotate);
    return hipGetLastError();
}

hipError_t xr_route_set_max_lds(size_t bytes) {
    const void* d2fns[6] = {reinterpret_cast<const void*>(&xr_route_kernel<true, XR_ZCH_DIAL2>),
                            reinterpret_cast<const void*>(&xr_step_queue_kernel<true, XR_ZCH_DIAL2>),
                            reinterpret_cast<const void*>(&xr_order_kernel<true, XR_ZCH_DIAL2>),
                            reinterpret_cast<const void*>(&xr_route_kernel<false, XR_ZCH_DIAL2>),
                            reinterpret_cast<const void*>(&xr_step_queue_kernel<false, XR_ZCH_DIAL2>),
                            reinterpret_cast<const void*>(&xr_order_kernel<false, XR_ZCH_DIAL2>)};
    for (int i = 0; i < 6; i++) {
        hipError_t e = hipFuncSetAttribute(d2fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
    }
    const void* d3fns[6] = {reinterpret_cast<const void*>(&xr_route_kernel<true, XR_ZCH_DIAL3>),
                            reinterpret_cast<const void*>(&xr_step_queue_kernel<true, XR_ZCH_DIAL3>),
                            reinterpret_cast<const void*>(&xr_order_kernel<true, XR_ZCH_DIAL3>),
                            reinterpret_cast<const void*>(&xr_route_kernel<true, XR_ZCH_DIAL3V2>),
                            reinterpret_cast<const void*>(&xr_step_queue_kernel<true, XR_ZCH_DIAL3V2>),
                            reinterpret_cast<const void*>(&xr_order_kernel<true, XR_ZCH_DIAL3V2>)};
    for (int i = 0; i < 6; i++) {
        hipError_t e = hipFuncSetAttribute(d3fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
    }
    const void* dfns[6] = {reinterpret_cast<const void*>(&xr_route_kernel<true, XR_ZCH_DIAL>),
                           reinterpret_cast<const void*>(&xr_step_queue_kernel<true, XR_ZCH_DIAL>),
                           reinterpret_cast<const void*>(&xr_order_kernel<true, XR_ZCH_DIAL>),
                           reinterpret_cast<const void*>(&xr_route_kernel<false, XR_ZCH_DIAL>),
                           reinterpret_cast<const void*>(&xr_step_queue_kernel<false, XR_ZCH_DIAL>),
                           reinterpret_cast<const void*>(&xr_order_kernel<false, XR_ZCH_DIAL>)};
    for (int i = 0; i < 6; i++) {
        hipError_t e = hipFuncSetAttribute(dfns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
    }
    const void* fns[6] = {reinterpret_cast<const void*>(&xr_route_kernel<true, 0>), reinterpret_cast<const void*>(&xr_route_kernel<true, 9>),
                          reinterpret_cast<const void*>(&xr_route_kernel<true, 12>), reinterpret_cast<const void*>(&xr_route_kernel<false, 0>),
                          reinterpret_cast<const void*>(&xr_route_kernel<false, 9>), reinterpret_cast<const void*>(&xr_route_kernel<false, 12>)};
    const void* qfns[6] = {reinterpret_cast<const void*>(&xr_step_queue_kernel<true, 0>), reinterpret_cast<const void*>(&xr_step_queue_kernel<true, 9>),
                           reinterpret_cast<const void*>(&xr_step_queue_kernel<true, 12>), reinterpret_cast<const void*>(&xr_step_queue_kernel<false, 0>),
                           reinterpret_cast<const void*>(&xr_step_queue_kernel<false, 9>), reinterpret_cast<const void*>(&xr_step_queue_kernel<false, 12>)};
    const void* ofns[6] = {reinterpret_cast<const void*>(&xr_order_kernel<true, 0>), reinterpret_cast<const void*>(&xr_order_kernel<true, 9>),
                           reinterpret_cast<const void*>(&xr_order_kernel<true, 12>), reinterpret_cast<const void*>(&xr_order_kernel<false, 0>),
                           reinterpret_cast<const void*>(&xr_order_kernel<false, 9>), reinterpret_cast<const void*>(&xr_order_kernel<false, 12>)};
    for (int i = 0; i < 6; i++) {
        hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(ofns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(qfns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// zch: 9 / 12 when every region of the batch has exactly that many layers, else 0
hipError_t xr_launch_route(const XrBatchDev* b, const int32_t* actions, int lds_dist, int zch, size_t lds_bytes,
                           int threads, hipStream_t st) {
    const dim3 g(b->env_count > 0 ? b->env_count : b->n_envs), t(threads);
    if (zch == XR_ZCH_DIAL3V2) {
        hipLaunchKernelGGL((xr_route_kernel<true, XR_ZCH_DIAL3V2>), g, t, lds_bytes, st, *b, actions);
    } else if (zch == XR_ZCH_DIAL3) {
        hipLaunchKernelGGL((xr_route_kernel<true, XR_ZCH_DIAL3>), g, t, lds_bytes, st, *b, actions);
    } else if (zch == XR_ZCH_DIAL2) {
        if (lds_dist) hipLaunchKernelGGL((xr_route_kernel<true, XR_ZCH_DIAL2>), g, t, lds_bytes, st, *b, actions);
        else hipLaunchKernelGGL((xr_route_kernel<false, XR_ZCH_DIAL2>), g, t, lds_bytes, st, *b, actions);
    } else if (zch == XR_ZCH_DIAL) {
        if (lds_dist) hipLaunchKernelGGL((xr_route_kernel<true, XR_ZCH_DIAL>), g, t, lds_bytes, st, *b, actions);
        else hipLaunchKernelGGL((xr_route_kernel<false, XR_ZCH_DIAL>), g, t, lds_bytes, st, *b, actions);
    } else if (lds_dist) {
        if (zch == 9) hipLaunchKernelGGL((xr_route_kernel<true, 9>), g, t, lds_bytes, st, *b, actions);
        else if (zch == 12) hipLaunchKernelGGL((xr_route_kernel<true, 12>), g, t, lds_bytes, st, *b, actions);
        else hipLaunchKernelGGL((xr_route_kernel<true, 0>), g, t, lds_bytes, st, *b, actions);
    } else {
        if (zch == 9) hipLaunchKernelGGL((xr_route_kernel<false, 9>), g, t, lds_bytes, st, *b, actions);
        else if (zch == 12) hipLaunchKernelGGL((xr_route_kernel<false, 12>), g, t, lds_bytes, st, *b, actions);
        else hipLaunchKernelGGL((xr_route_kernel<false, 0>), g, t, lds_bytes, st, *b, actions);
    }
    return hipGetLastError();
}

hipError_t xr_launch_order(const XrBatchDev* b, const int32_t* orders, int stride, int32_t* net_stats, int lds_dist, int zch,
                           size_t lds_bytes, int threads, hipStream_t st) {
    const dim3 g(b->n_envs), t(threads);
    if (zch == XR_ZCH_DIAL3V2) {
        hipLaunchKernelGGL((xr_order_kernel<true, XR_ZCH_DIAL3V2>), g, t, lds_bytes, st, *b, orders, stride, net_stats);
    } else if (zch == XR_ZCH_DIAL3) {
        hipLaunchKernelGGL((xr_order_kernel<true, XR_ZCH_DIAL3>), g, t, lds_bytes, st, *b, orders, stride, net_stats);
    } else if (zch == XR_ZCH_DIAL2) {
        if (lds_dist) hipLaunchKernelGGL((xr_order_kernel<true, XR_ZCH_DIAL2>), g, t, lds_bytes, st, *b, orders, stride, net_stats);
        else hipLaunchKernelGGL((xr_order_kernel<false, XR_ZCH_DIAL2>), g, t, lds_bytes, st, *b, orders, stride, net_stats);
    } else if (zch == XR_ZCH_DIAL) {
        if (lds_dist) hipLaunchKernelGGL((xr_order_kernel<true, XR_ZCH_DIAL>), g, t, lds_bytes, st, *b, orders, stride, net_stats);
        else hipLaunchKernelGGL((xr_order_kernel<false, XR_ZCH_DIAL>), g, t, lds_bytes, st, *b, orders, stride, net_stats);
    } else if (lds_dist) {
        if (zch == 9) hipLaunchKernelGGL((xr_order_kernel<true, 9>), g, t, lds_bytes, st, *b, orders, stride, net_stats);
        else if (zch == 12) hipLaunchKernelGGL((xr_order_kernel<true, 12>), g, t, lds_bytes, st, *b, orders, stride, net_stats);
        else hipLaunchKernelGGL((xr_order_kernel<true, 0>), g, t, lds_bytes, st, *b, orders, stride, net_stats);
    } else {
        if (zch == 9) hipLaunchKernelGGL((xr_order_kernel<false, 9>), g, t, lds_bytes, st, *b, orders, stride, net_stats);
        else if (zch == 12) hipLaunchKernelGGL((xr_order_kernel<false, 12>), g, t, lds_bytes, st, *b, orders, stride, net_stats);
        else hipLaunchKernelGGL((xr_order_kernel<false, 0>), g, t, lds_bytes, st, *b, orders, stride, net_stats);
    }
    return hipGetLastError();
}

// resident workgroups per CU of the step kernel as the runtime would place it, and its static LDS
hipError_t xr_route_occupancy(int lds_dist, int zch, size_t lds_bytes, int threads, int* wg_per_cu, size_t* static_lds) {
    const void* fn = zch == XR_ZCH_DIAL3V2 ? reinterpret_cast<const void*>(&xr_route_kernel<true, XR_ZCH_DIAL3V2>)
                     : zch == XR_ZCH_DIAL3 ? reinterpret_cast<const void*>(&xr_route_kernel<true, XR_ZCH_DIAL3>)
                     : zch == XR_ZCH_DIAL2 ? (lds_dist ? reinterpret_cast<const void*>(&xr_route_kernel<true, XR_ZCH_DIAL2>)
                                                     : reinterpret_cast<const void*>(&xr_route_kernel<false, XR_ZCH_DIAL2>))
                     : zch == XR_ZCH_DIAL ? (lds_dist ? reinterpret_cast<const void*>(&xr_route_kernel<true, XR_ZCH_DIAL>)
                                                    : reinterpret_cast<const void*>(&xr_route_kernel<false, XR_ZCH_DIAL>))
                     : lds_dist ? (zch == 9 ? reinterpret_cast<const void*>(&xr_route_kernel<true, 9>)
                                 : zch == 12 ? reinterpret_cast<const void*>(&xr_route_kernel<true, 12>)
                                             : reinterpret_cast<const void*>(&xr_route_kernel<true, 0>))
                              : (zch == 9 ? reinterpret_cast<const void*>(&xr_route_kernel<false, 9>)
                                 : zch == 12 ? reinterpret_cast<const void*>(&xr_route_kernel<false, 12>)
                                             : reinterpret_cast<const void*>(&xr_route_kernel<false, 0>));
    hipFuncAttributes attr;
    hipError_t e = hipFuncGetAttributes(&attr, fn);
    if (e != hipSuccess) return e;
    *static_lds = attr.sharedSizeBytes;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(wg_per_cu, fn, threads, lds_bytes);
}

hipError_t xr_launch_step_queue(const XrBatchDev* b, const int32_t* actions, int lds_dist, int zch, size_t lds_bytes,
                                int threads, int blocks, hipStream_t st) {
    const dim3 g(blocks), t(threads);
    if (zch == XR_ZCH_DIAL3V2) {
        hipLaunchKernelGGL((xr_step_queue_kernel<true, XR_ZCH_DIAL3V2>), g, t, lds_bytes, st, *b, actions);
    } else if (zch == XR_ZCH_DIAL3) {
        hipLaunchKernelGGL((xr_step_queue_kernel<true, XR_ZCH_DIAL3>), g, t, lds_bytes, st, *b, actions);
    } else if (zch == XR_ZCH_DIAL2) {
        if (lds_dist) hipLaunchKernelGGL((xr_step_queue_kernel<true, XR_ZCH_DIAL2>), g, t, lds_bytes, st, *b, actions);
        else hipLaunchKernelGGL((xr_step_queue_kernel<false, XR_ZCH_DIAL2>), g, t, lds_bytes, st, *b, actions);
    } else if (zch == XR_ZCH_DIAL) {
        if (lds_dist) hipLaunchKernelGGL((xr_step_queue_kernel<true, XR_ZCH_DIAL>), g, t, lds_bytes, st, *b, actions);
        else hipLaunchKernelGGL((xr_step_queue_kernel<false, XR_ZCH_DIAL>), g, t, lds_bytes, st, *b, actions);
    } else if (lds_dist) {
        if (zch == 9) hipLaunchKernelGGL((xr_step_queue_kernel<true, 9>), g, t, lds_bytes, st, *b, actions);
        else if (zch == 12) hipLaunchKernelGGL((xr_step_queue_kernel<true, 12>), g, t, lds_bytes, st, *b, actions);
        else hipLaunchKernelGGL((xr_step_queue_kernel<true, 0>), g, t, lds_bytes, st, *b, actions);
    } else {
        if (zch == 9) hipLaunchKernelGGL((xr_step_queue_kernel<false, 9>), g, t, lds_bytes, st, *b, actions);
        else if (zch == 12) hipLaunchKernelGGL((xr_step_queue_kernel<false, 12>), g, t, lds_bytes, st, *b, actions);
        else hipLaunchKernelGGL((xr_step_queue_kernel<false, 0>), g, t, lds_bytes, st, *b, actions);
    }
    return hipGetLastError();
}

hipError_t xr_launch_route_order(const XrBatchDev* b, const int32_t* actions, int32_t* order, hipStream_t st) {
    hipLaunchKernelGGL(xr_route_order_kernel, dim3(1), dim3(1024), 0, st, *b, actions, order);
    return hipGetLastError();
}

// b->queue: this call's counters (zero: the previous call's plan zeroed them); next_queue: the other bank.  order != null: the route
// tasks' longest-first order is wanted too — by the same launch when the batch fits one workgroup (*order_done = 1), else the
// caller launches xr_route_order_kernel.
hipError_t xr_launch_plan(const XrBatchDev* b, const int32_t* actions, uint32_t* next_queue, int32_t* order, int* order_done, hipStream_t st) {
    if (order_done) *order_done = 0;
    if (order && b->n_envs <= 1024) {
        hipLaunchKernelGGL(xr_plan_kernel<1024>, dim3(1), dim3(1024), 0, st, *b, actions, next_queue, order);
        if (order_done) *order_done = 1;
    } else {
        hipLaunchKernelGGL(xr_plan_kernel<256>, dim3((b->n_envs + 255) / 256), dim3(256), 0, st, *b, actions, next_queue, (int32_t*)nullptr);
    }
    return hipGetLastError();
}

hipError_t xr_launch_netplanes(const XrBatchDev* b, int blocks, int aligned, hipStream_t st) {
    if (aligned) hipLaunchKernelGGL(xr_netplane_kernel, dim3(blocks), dim3(256), 0, st, *b);
    else hipLaunchKernelGGL(xr_netplane_stream_kernel, dim3(blocks), dim3(256), (size_t)(((b->n_max + 15) / 16 + 2) * 4), st, *b);
    return hipGetLastError();
}

hipError_t xr_launch_random_actions(const XrBatchDev* b, int32_t* actions, uint64_t seed, hipStream_t st) {
    hipLaunchKernelGGL(xr_random_action_kernel, dim3((b->n_envs + 255) / 256), dim3(256), 0, st, *b, actions, seed);
    return hipGetLastError();
}

hipError_t xr_launch_obs(const XrBatchDev* b, float* out, int64_t env_stride, int env_lo, int env_hi, int n_max_nodes,
                         int vec4, hipStream_t st) {
    const int n_env = env_hi - env_lo;
    if (n_env <= 0) return hipSuccess;
    const size_t lds = (size_t)(b->legal_words * 64 + b->legal_words + 1) * sizeof(int);
    // gridDim.y <= 65535: slice the env range
    for (int lo = env_lo; lo < env_hi; lo += 32768) {
        const int cnt = (env_hi - lo) < 32768 ? (env_hi - lo) : 32768;
        float* o = out + (int64_t)(lo - env_lo) * env_stride;
        if (vec4 == 1) {
            const int chunks = (n_max_nodes + 1023) / 1024;
            hipLaunchKernelGGL(xr_obs_kernel<4>, dim3(chunks, cnt), dim3(256), lds, st, *b, o, env_stride, lo);
        } else if (vec4 == 2) {
            const size_t lds2 = (size_t)(b->legal_words * 64 + ((b->legal_words + 1 + 3) & ~3)) * sizeof(int) + (size_t)b->n_max * 2;
            hipLaunchKernelGGL(xr_obs_stream_kernel, dim3(cnt), dim3(256), lds2, st, *b, o, env_stride, lo);
        } else {
            const int chunks = (n_max_nodes + 255) / 256;
            hipLaunchKernelGGL(xr_obs_kernel<1>, dim3(chunks, cnt), dim3(256), lds, st, *b, o, env_stride, lo);
        }
    }
    return hipGetLastError();
}

hipError_t xr_launch_unit_helpers(const XrBatchDev* b, int blocks, hipStream_t st) {
    hipLaunchKernelGGL(xr_unit_helper_kernel, dim3(blocks), dim3(256), 0, st, *b);
    return hipGetLastError();
}

hipError_t xr_launch_netplanes_pairs(const XrBatchDev* b, const int32_t* pair_region, const int32_t* pair_net, int n_pairs,
                                     float* out, int64_t pair_stride, int vec4, hipStream_t st) {
    if (n_pairs <= 0) return hipSuccess;
    hipLaunchKernelGGL(xr_netplanes_pairs_kernel, dim3(n_pairs), dim3(256), 0, st, *b, pair_region, pair_net, out, pair_stride, vec4);
    return hipGetLastError();
}

hipError_t xr_launch_guide_masks(const XrBatchDev* b, uint8_t* masks, int k_max, hipStream_t st) {
    if (b->n_regions <= 0 || k_max <= 0) return hipSuccess;
    // (gridDim.y holds at most 65535 workgroups: the regions go in chunks)
    for (int r0 = 0; r0 < b->n_regions; r0 += 32768) {
        hipLaunchKernelGGL(xr_guide_mask_kernel, dim3(k_max, (b->n_regions - r0) < 32768 ? (b->n_regions - r0) : 32768), dim3(256), 0, st, *b, masks, r0);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t xr_launch_ingest_state(const XrBatchDev* b, const int16_t* owner_in, const uint64_t* legal_in, const int32_t* cum_in, hipStream_t st) {
    if (b->n_envs <= 0) return hipSuccess;
    hipLaunchKernelGGL(xr_ingest_state_kernel, dim3(b->n_envs), dim3(256), 0, st, *b, owner_in, legal_in, cum_in);
    return hipGetLastError();
}

hipError_t xr_launch_pack_state(const XrBatchDev* b, uint8_t* rows, int64_t row_bytes, int region_base, hipStream_t st) {
    if (b->n_envs <= 0) return hipSuccess;
    hipLaunchKernelGGL(xr_pack_state_kernel, dim3(b->n_envs), dim3(256), 0, st, *b, rows, row_bytes, region_base);
    return hipGetLastError();
}

hipError_t xr_launch_expand_state(const XrBatchDev* b, const uint8_t* rows, int64_t row_bytes, int n_rows, float* head, int64_t head_stride,
                                  int32_t* nlegal_out, int32_t* region_out, int vec4, hipStream_t st) {
    if (n_rows <= 0) return hipSuccess;
    const size_t lds = (size_t)(b->legal_words * 64 + b->legal_words + 1) * sizeof(int);
    hipLaunchKernelGGL(xr_expand_state_kernel, dim3(n_rows), dim3(256), lds, st, *b, rows, row_bytes, head, head_stride, nlegal_out, region_out, vec4);
    return hipGetLastError();
}

hipError_t xr_launch_obs_records(const uint32_t* rec, int X, int Y, int Z, const int32_t* nets, int K, float* out,
                                 int vec4, hipStream_t st) {
    const int N = X * Y * Z;
    if (N <= 0) return hipSuccess;
    const size_t lds = (size_t)(K > 0 ? K : 1) * sizeof(int);
    if (vec4)
        hipLaunchKernelGGL(xr_obs_records_kernel<4>, dim3((N + 1023) / 1024), dim3(256), lds, st, rec, X, Y, Z, nets, K, out);
    else
        hipLaunchKernelGGL(xr_obs_records_kernel<1>, dim3((N + 255) / 256), dim3(256), lds, st, rec, X, Y, Z, nets, K, out);
    return hipGetLastError();
}

}  // extern "C"
