// xr_kernels.hip — gfx950 (MI355X, wave64) kernels of the env hot path.
//
//   xr_ingest_kernel      packed node records -> compact state (node_net, owner0)      HBM sweep
//   xr_reset_kernel       Game.reset bookkeeping for masked envs                        HBM copy
//   xr_route_kernel       Game.step: grid build -> XR-Maze v1 maze route -> claim path -> metrics
//                         one workgroup per env; distance field resident in LDS (<= ~32k nodes)
//                         or in a per-env HBM scratch (larger regions)
//   xr_obs_kernel         build_3Dgrid: compact state -> fp32 [2+7K, Z, Y, X] observation, streaming
//   xr_random_action_kernel
//
// Integer / index work throughout: no MFMA.  What matters here is coalescing (every sweep is
// unit-stride in the reference's own flat order f=(x*Y+y)*Z+z), LDS residency of the distance
// field, and wave64-wide relaxation.
#include <hip/hip_runtime.h>
#include <stdint.h>

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
// Relaxation = alternating line sweeps ("fast sweeping" on the routing graph):
//   phase T  thread <-> track (one preferred-direction line of one layer): forward and backward
//            Gauss-Seidel pass along the whole line, values staged in registers in chunks, so a
//            distance travels any straight run in ONE pass;
//   phase V  thread <-> (x,y) column: up and down pass over the via chain.
// A path with b bends/vias is resolved in ~b iterations instead of ~hops sweeps.  Convergence:
// __syncthreads_or of "I lowered a distance" over one full T+V iteration.
// Pruning: nothing above `bound` (= best distance of any unconnected target so far) is written; every
// node with true distance <= the final target distance still gets its exact value (induction along
// its shortest path), so the target choice and the back-trace are unchanged.
// Later pins of the same net re-use the field: every value is still an upper bound once the new path
// nodes are set to 0, so relaxation continues instead of restarting.
//
// LDS layout of the field: index l = x*SX + y*SY + z with SY = Z|1, SX = (Y*SY)|1 (odd strides):
// lanes of a wave hold consecutive tracks / columns, i.e. consecutive y or consecutive x, so every
// wave access has an odd word stride and is bank-conflict free.
// ------------------------------------------------------------------------------------------------
#define XR_CH 8          // track chunk held in registers
#define XR_CLAIM 4u      // cls bit: node claimed by the back-trace in progress

__device__ __forceinline__ uint32_t xr_pen_of(uint32_t c, uint32_t pen) { return (c & 3u) == XR_CLS_PEN ? pen : 0u; }

// one Gauss-Seidel pass along a line of L nodes: l(i) = base + i*stride, coordinate c[i]
template <bool FWD>
__device__ __forceinline__ int xr_line_pass(uint32_t* __restrict__ dist, const uint8_t* __restrict__ cls,
                                            const int32_t* __restrict__ co, int base, int stride, int L,
                                            uint32_t pen, uint32_t bound) {
    int changed = 0;
    uint32_t prev = XR_INF;
    int prevc = 0;
    for (int i0 = 0; i0 < L; i0 += XR_CH) {
        uint32_t d[XR_CH];
        uint32_t c[XR_CH];
        int cc[XR_CH];
#pragma unroll
        for (int j = 0; j < XR_CH; j++) {
            const int i = FWD ? (i0 + j) : (L - 1 - i0 - j);
            const bool in = FWD ? (i < L) : (i >= 0);
            const int l = base + i * stride;
            d[j] = in ? dist[l] : XR_INF;
            c[j] = in ? (uint32_t)cls[l] : (uint32_t)XR_CLS_BLOCK;
            cc[j] = in ? co[i] : 0;
        }
#pragma unroll
        for (int j = 0; j < XR_CH; j++) {
            const int i = FWD ? (i0 + j) : (L - 1 - i0 - j);
            const bool in = FWD ? (i < L) : (i >= 0);
            if (in) {
                if ((c[j] & 3u) == XR_CLS_BLOCK) {
                    prev = XR_INF;
                } else {
                    if (prev != XR_INF) {
                        const uint32_t len = (uint32_t)(FWD ? (cc[j] - prevc) : (prevc - cc[j]));
                        const uint32_t cand = prev + len + xr_pen_of(c[j], pen);
                        if (cand < d[j] && cand <= bound) {
                            d[j] = cand;
                            dist[base + i * stride] = cand;
                            changed = 1;
                        }
                    }
                    prev = d[j];
                }
                prevc = cc[j];
            }
        }
    }
    return changed;
}

template <bool LDS_DIST>
__global__ void xr_route_kernel(XrBatchDev b, const int32_t* __restrict__ actions) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_ap_l[XR_MAX_AP_PER_NET];          // padded LDS index of each access point
    __shared__ short s_ap_pin[XR_MAX_AP_PER_NET];
    __shared__ unsigned char s_ap_conn[XR_MAX_AP_PER_NET];
    __shared__ unsigned char s_hl[XR_MAX_LAYERS], s_vl[XR_MAX_LAYERS];
    __shared__ int s_remaining, s_target_i;
    __shared__ uint32_t s_bound;

    const int e = blockIdx.x;
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
        }
        return;
    }

    const XrRegionDev R = b.regions[b.env_region[e]];
    const int a = actions[e];
    bool valid = (a >= 1 && a <= R.n_nets);
    if (valid) valid = (b.legal[(int64_t)e * b.legal_words + ((a - 1) >> 6)] >> ((a - 1) & 63)) & 1ULL;
    if (!valid) {   // the reference never checks this client-side; here: flagged no-op
        if (tid == 0) {
            b.status[e] = XR_ENV_BAD_ACTION;
            b.delta[3 * e] = 0; b.delta[3 * e + 1] = 0; b.delta[3 * e + 2] = 0;
            b.reward[e] = -0.0; b.path_len[e] = 0; b.sweeps[e] = 0;
        }
        return;
    }

    const int X = R.X, Y = R.Y, Z = R.Z;
    const int SY = Z | 1, SX = (Y * SY) | 1;      // odd strides
    const int ncol = X * Y;
    const int16_t* __restrict__ node_net = b.rg_node_net + R.node_off;
    int16_t* __restrict__ owner = b.owner + (int64_t)e * b.n_max;

    // carve: dist u32[n_lds] | cls u8[n_lds] | xs i32[X] | ys i32[Y]
    uint32_t* dist;
    uint8_t* cls;
    int32_t* s_xs;
    if (LDS_DIST) {
        dist = reinterpret_cast<uint32_t*>(smem);
        cls = reinterpret_cast<uint8_t*>(smem + (size_t)b.n_lds * 4);
        s_xs = reinterpret_cast<int32_t*>(smem + (size_t)b.n_lds * 5);   // n_lds % 8 == 0 -> aligned
    } else {
        dist = b.dist_scratch + (int64_t)e * b.n_lds;
        cls = b.cls_scratch + (int64_t)e * b.n_lds;
        s_xs = reinterpret_cast<int32_t*>(smem);
    }
    int32_t* s_ys = s_xs + X;

    // ---- grid build: cost class of every node for THIS net, distance field = INF ------------------
    for (int col = tid; col < ncol; col += nthr) {
        const int x = col / Y, y = col - x * Y;
        const int g0 = col * Z, l0 = x * SX + y * SY;
        for (int z = 0; z < Z; z++) {
            const int nn = node_net[g0 + z];
            const int ow = owner[g0 + z];
            uint8_t c = XR_CLS_FREE;
            if (nn == -1) c = XR_CLS_BLOCK;
            else if ((ow != 0 && ow != a) || (nn > 0 && nn != a)) c = XR_CLS_PEN;
            cls[l0 + z] = c;
            dist[l0 + z] = XR_INF;
        }
    }
    for (int i = tid; i < X; i += nthr) s_xs[i] = b.coords[R.xs_off + i];
    for (int i = tid; i < Y; i += nthr) s_ys[i] = b.coords[R.ys_off + i];
    const int ap_lo = b.net_csr[R.net_off + a], ap_hi = b.net_csr[R.net_off + a + 1];
    const int nap = ap_hi - ap_lo;    // 1 <= nap <= XR_MAX_AP_PER_NET (checked at load)
    for (int i = tid; i < nap; i += nthr) {
        const int f = b.ap_node[R.ap_off + ap_lo + i];
        const int z = f % Z, y = (f / Z) % Y, x = f / (Y * Z);
        s_ap_l[i] = x * SX + y * SY + z;
        s_ap_pin[i] = b.ap_pin[R.ap_off + ap_lo + i];
        s_ap_conn[i] = 0;
    }
    if (tid == 0) {            // layer tables: horizontal layers carry x-tracks, vertical layers y-tracks
        int nh = 0, nv = 0;
        for (int z = 0; z < Z; z++) {
            if ((R.ldir_mask >> z) & 1u) s_vl[nv++] = (unsigned char)z; else s_hl[nh++] = (unsigned char)z;
        }
    }
    __syncthreads();

    // ---- component = all access points of the lowest pin id -------------------------------------
    if (tid == 0) {
        int first = 0x7FFFFFFF;
        for (int i = 0; i < nap; i++) first = min(first, (int)s_ap_pin[i]);
        int npins = 0;
        for (int i = 0; i < nap; i++) {
            bool seen = false;
            for (int j = 0; j < i; j++) if (s_ap_pin[j] == s_ap_pin[i]) { seen = true; break; }
            npins += !seen;
            if (s_ap_pin[i] == first) { s_ap_conn[i] = 1; dist[s_ap_l[i]] = 0; }
        }
        s_remaining = npins - 1;
        s_bound = XR_INF;
    }
    __syncthreads();

    const uint32_t via = (uint32_t)b.via_cost;
    const uint32_t pen = (uint32_t)b.pen_cost;
    const int nv_layers = __popc(R.ldir_mask & (Z >= 32 ? 0xFFFFFFFFu : ((1u << Z) - 1u)));
    const int nh_layers = Z - nv_layers;
    const int tracks_h = nh_layers * Y;            // lines along x, one per (y, horizontal layer)
    const int ntracks = tracks_h + nv_layers * X;  // + lines along y, one per (x, vertical layer)
    int d_vio = 0, d_wl = 0, d_via = 0, plen = 0, status = XR_ENV_OK, nsweeps = 0;   // thread 0 only
    uint64_t h = (tid == 0) ? b.hash[e] : 0;
    int32_t* __restrict__ path = b.path + (int64_t)e * b.path_cap;

    while (s_remaining > 0) {
        // ---- relax to the (pruned) fixpoint ----------------------------------------------------
        int any;
        do {
            // tighten the bound from the targets' current distances
            for (int i = tid; i < nap; i += nthr)
                if (!s_ap_conn[i]) { const uint32_t dv = dist[s_ap_l[i]]; if (dv != XR_INF) atomicMin(&s_bound, dv); }
            __syncthreads();
            const uint32_t bound = s_bound;
            int changed = 0;
            // phase T: line sweeps
            for (int t = tid; t < ntracks; t += nthr) {
                int base, stride, L;
                const int32_t* co;
                if (t < tracks_h) {
                    const int zi = t / Y, y = t - zi * Y;
                    base = y * SY + s_hl[zi]; stride = SX; L = X; co = s_xs;
                } else {
                    const int u = t - tracks_h;
                    const int zi = u / X, x = u - zi * X;
                    base = x * SX + s_vl[zi]; stride = SY; L = Y; co = s_ys;
                }
                changed |= xr_line_pass<true>(dist, cls, co, base, stride, L, pen, bound);
                changed |= xr_line_pass<false>(dist, cls, co, base, stride, L, pen, bound);
            }
            __syncthreads();
            // phase V: via chains
            for (int col = tid; col < ncol; col += nthr) {
                const int x = col / Y, y = col - x * Y;
                const int l0 = x * SX + y * SY;
                uint32_t prev = XR_INF;
                for (int z = 0; z < Z; z++) {
                    const uint32_t c = cls[l0 + z];
                    if ((c & 3u) == XR_CLS_BLOCK) { prev = XR_INF; continue; }
                    uint32_t cur = dist[l0 + z];
                    if (prev != XR_INF) {
                        const uint32_t cand = prev + via + xr_pen_of(c, pen);
                        if (cand < cur && cand <= bound) { cur = cand; dist[l0 + z] = cand; changed = 1; }
                    }
                    prev = cur;
                }
                prev = XR_INF;
                for (int z = Z - 1; z >= 0; z--) {
                    const uint32_t c = cls[l0 + z];
                    if ((c & 3u) == XR_CLS_BLOCK) { prev = XR_INF; continue; }
                    uint32_t cur = dist[l0 + z];
                    if (prev != XR_INF) {
                        const uint32_t cand = prev + via + xr_pen_of(c, pen);
                        if (cand < cur && cand <= bound) { cur = cand; dist[l0 + z] = cand; changed = 1; }
                    }
                    prev = cur;
                }
            }
            any = __syncthreads_or(changed);
            nsweeps++;
        } while (any);

        // ---- nearest access point of an unconnected pin; ties -> lowest flat index (wave 0) --------
        if (tid < 64) {
            unsigned long long best = ~0ULL;
            for (int i = tid; i < nap; i += 64) {
                if (s_ap_conn[i]) continue;
                const uint32_t dv = dist[s_ap_l[i]];
                if (dv == XR_INF) continue;
                // (distance, padded index): padded index order == flat index order
                const unsigned long long key = ((unsigned long long)dv << 32) | (unsigned)s_ap_l[i];
                best = key < best ? key : best;
            }
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long o = __shfl_xor(best, off);
                best = o < best ? o : best;
            }
            int best_i = -1;
            if (best != ~0ULL) {                  // AP slot holding that node (node ids are unique per net)
                const int best_l = (int)(best & 0xFFFFFFFFu);
                for (int i0 = 0; i0 < nap && best_i < 0; i0 += 64) {
                    const int i = i0 + tid;
                    const unsigned long long m = __ballot(i < nap && s_ap_l[i] == best_l);
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
                // Lanes 0..5 test one direction each; ballot + ffs picks the first match.
                int v = s_ap_l[best_i];
                uint32_t dv = dist[v];
                uint32_t cv = cls[v];
                while (dv > 0) {
                    const int x = v / SX, r = v - x * SX, y = r / SY, z = r - y * SY;
                    const uint32_t need = dv - xr_pen_of(cv, pen);   // dist[u] + len must equal this
                    const bool vert = (R.ldir_mask >> z) & 1u;
                    int u = -1;
                    uint32_t len = 0;
                    switch (tid) {
                    case 0: if (!vert && x + 1 < X) { u = v + SX; len = (uint32_t)(s_xs[x + 1] - s_xs[x]); } break;   // E
                    case 1: if (vert && y > 0)      { u = v - SY; len = (uint32_t)(s_ys[y] - s_ys[y - 1]); } break;   // S
                    case 2: if (!vert && x > 0)     { u = v - SX; len = (uint32_t)(s_xs[x] - s_xs[x - 1]); } break;   // W
                    case 3: if (vert && y + 1 < Y)  { u = v + SY; len = (uint32_t)(s_ys[y + 1] - s_ys[y]); } break;   // N
                    case 4: if (z + 1 < Z)          { u = v + 1; len = via; } break;                                   // U
                    case 5: if (z > 0)              { u = v - 1; len = via; } break;                                   // D
                    default: break;
                    }
                    uint32_t du = XR_INF, cu = XR_CLS_BLOCK;
                    bool ok = false;
                    if (u >= 0) {
                        cu = cls[u];
                        du = dist[u];
                        ok = ((cu & 3u) != XR_CLS_BLOCK) && du != XR_INF && du + len == need;
                    }
                    const unsigned long long m = __ballot(ok);
                    if (m == 0) { if (tid == 0) status |= 0x100; break; }     // inconsistent field: cannot happen
                    const int src = __ffsll((long long)m) - 1;
                    const int pu = __shfl(u, src);
                    const uint32_t pdu = __shfl(du, src), pcu = __shfl(cu, src), pl = __shfl(len, src);
                    if (tid == 0) {                 // claim v
                        const int f = (x * Y + y) * Z + z;
                        if ((cv & 3u) == XR_CLS_PEN) d_vio += 1;
                        if (owner[f] == 0) owner[f] = (int16_t)a;
                        cls[v] = (uint8_t)(cv | XR_CLAIM);     // dist[v] = 0 after the trace (the trace reads the field)
                        if (plen < b.path_cap) path[plen] = f;
                        plen++;
                        fnv_mix(h, (uint32_t)f);
                        if (src >= 4) d_via += 1; else d_wl += (int)pl;
                    }
                    v = pu; dv = pdu; cv = pcu;
                }
                if (tid == 0 && (status & 0x100)) {
                    s_remaining = 0;              // never taken on a consistent field; avoids spinning
                } else if (tid == 0) {
                    // terminal node of the component: claimed (and recorded) only if nobody holds it yet
                    const int x = v / SX, r = v - x * SX, y = r / SY, z = r - y * SY;
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
        // path nodes and the reached pin's access points become sources of the next search
        {
            const int ti = s_target_i;
            if (ti >= 0) {
                const short pin = s_ap_pin[ti];
                for (int i = tid; i < nap; i += nthr)
                    if (s_ap_pin[i] == pin) { s_ap_conn[i] = 1; dist[s_ap_l[i]] = 0; }
                for (int col = tid; col < ncol; col += nthr) {
                    const int x = col / Y, y = col - x * Y;
                    const int l0 = x * SX + y * SY;
                    for (int z = 0; z < Z; z++) {
                        const uint32_t c = cls[l0 + z];
                        if (c & XR_CLAIM) { cls[l0 + z] = (uint8_t)(c & 3u); dist[l0 + z] = 0; }
                    }
                }
            }
        }
        __syncthreads();
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
    obst = (n == -1 || s.used(f)) ? 1.f : 0.f;
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
template <class Src, int VEC>
__device__ __forceinline__ void xr_obs_write(const Src& s, int X, int Y, int Z, int N, const int* s_ids, int K,
                                             float* __restrict__ out, int chunk_base) {
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
        *reinterpret_cast<float4*>(p) = v;
        p += N;
        v.x = (f0 + 0 < K) ? (float)s_ids[f0 + 0] : 0.f;
        v.y = (f0 + 1 < K) ? (float)s_ids[f0 + 1] : 0.f;
        v.z = (f0 + 2 < K) ? (float)s_ids[f0 + 2] : 0.f;
        v.w = (f0 + 3 < K) ? (float)s_ids[f0 + 3] : 0.f;
        *reinterpret_cast<float4*>(p) = v;
        p += N;
        bool anyap = false;
#pragma unroll
        for (int j = 0; j < VEC; j++) anyap |= (apnet[j] != 0);
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = 0; i < K; i++) {
            float4 m = zero, ma = zero;
            if (anyap) {
                const int id = s_ids[i];
                m.x = (apnet[0] == id) ? 1.f : 0.f; m.y = (apnet[1] == id) ? 1.f : 0.f;
                m.z = (apnet[2] == id) ? 1.f : 0.f; m.w = (apnet[3] == id) ? 1.f : 0.f;
                ma.x = (adj[0] ? m.x : 0.f); ma.y = (adj[1] ? m.y : 0.f);
                ma.z = (adj[2] ? m.z : 0.f); ma.w = (adj[3] ? m.w : 0.f);
            }
            *reinterpret_cast<float4*>(p) = m;
            p += N;
#pragma unroll
            for (int c = 0; c < 6; c++) { *reinterpret_cast<float4*>(p) = ma; p += N; }
        }
    } else {
        p[0] = obst[0];
        p += N;
        p[0] = (f0 < K) ? (float)s_ids[f0] : 0.f;
        p += N;
        for (int i = 0; i < K; i++) {
            const float m = (apnet[0] != 0 && apnet[0] == s_ids[i]) ? 1.f : 0.f;
            const float ma = adj[0] ? m : 0.f;
            p[0] = m;
            p += N;
#pragma unroll
            for (int c = 0; c < 6; c++) { p[0] = ma; p += N; }
        }
    }
}

// legal bitmask -> ascending id list in LDS (== sorted(list(netSet)), build_3Dgrid.py:177)
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
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&xr_route_kernel<true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

hipError_t xr_launch_route(const XrBatchDev* b, const int32_t* actions, int lds_dist, size_t lds_bytes, int threads,
                           hipStream_t st) {
    if (lds_dist)
        hipLaunchKernelGGL(xr_route_kernel<true>, dim3(b->n_envs), dim3(threads), lds_bytes, st, *b, actions);
    else
        hipLaunchKernelGGL(xr_route_kernel<false>, dim3(b->n_envs), dim3(threads), lds_bytes, st, *b, actions);
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
        if (vec4) {
            const int chunks = (n_max_nodes + 1023) / 1024;
            hipLaunchKernelGGL(xr_obs_kernel<4>, dim3(chunks, cnt), dim3(256), lds, st, *b, o, env_stride, lo);
        } else {
            const int chunks = (n_max_nodes + 255) / 256;
            hipLaunchKernelGGL(xr_obs_kernel<1>, dim3(chunks, cnt), dim3(256), lds, st, *b, o, env_stride, lo);
        }
    }
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
