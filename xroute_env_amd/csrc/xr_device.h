// xr_device.h — structures shared by the host side (xr_batch.cpp) and the gfx950 kernels
// (xr_kernels.hip).  Internal: the public boundary is include/xroute_hip.h.
#pragma once
#include <stdint.h>

#define XR_INF 0xFFFFFFFFu
#define XR_CLS_FREE 0
#define XR_CLS_PEN 1
#define XR_CLS_BLOCK 2
// work-list capacities of the frontier router's HBM-scratch form (xr_dial.h); capacity never affects results
#ifndef XR_BIG_CA
#define XR_BIG_CA 2048         // active words per chunk
#endif
#ifndef XR_BIG_CN
#define XR_BIG_CN 4096         // nodes per chunk
#endif
#ifndef XR_BIG_CE
#define XR_BIG_CE 2048         // (node, distance) pairs per expansion pass
#endif
#define XR_BIG_MAXG 1024       // groups (1024 nodes each): regions up to 1 M nodes
#define XR_MAX_AP_PER_NET 128   // access points of one net staged in LDS by the route kernel
// round 3's LDS form of the frontier router (xr_dial3.h): its node list (path chunks) and its LDS footprint
#ifndef XR3_TMP
#define XR3_TMP 512           // node list (u16): path nodes of a back-trace, flushed in chunks
#endif
#define XR3_LDS_BYTES(n_max, x_max, y_max) ((size_t)(n_max) * 4 + 3 * ((size_t)(n_max) / 32 + 1) * 4 + \
                                            ((size_t)(x_max) + (size_t)(y_max) + 4) * 4 + (size_t)XR3_TMP * 2 + 16)
#define XR3_STEP_LIMIT (1ll << 20)     // one edge + every penalty it can carry (the word's x32 fixed point: cap x 32 + step x 32 < 2^32)
#define XR3_EXTENT_LIMIT (1ll << 25)   // widest track span of a region (x32 coordinate tables; heuristic sums of two spans + 31 via costs in 31 bits)

// One region (static after xr_batch_load_regions). Node arrays are in the reference observation's
// flat order f = (x*Y + y)*Z + z.
struct XrRegionDev {
    int32_t X, Y, Z, N;
    int32_t n_nets;        // highest 1-based net id that may appear
    int32_t nlegal0;       // nets with at least one access point
    int32_t m0[3];         // cumulative metrics of the initial Request
    uint32_t ldir_mask;    // bit z set: layer z is vertical (y moves); clear: horizontal (x moves)
    int32_t xs_off, ys_off;  // into coords[]
    int32_t net_off;       // into net_csr[]: APs of net n are [csr[n], csr[n+1]) (+ ap_off)
    int32_t ap_off;        // into ap_node[] / ap_pin[]
    int64_t node_off;      // into rg_rec / rg_node_net / rg_owner0
    int64_t legal0_off;    // into legal0[] (uint64 words, legal_words per region)
    // bucketed-frontier router (xr_dial.h)
    uint32_t w_min;        // smallest edge length of the region graph: min(x pitches, y pitches, via cost)
    uint32_t magic_yz;     // floor(2^32 / (Y*Z)) and floor(2^32 / Z) (0xFFFFFFFF for a divisor of 1): flat index -> (x, y, z)
    uint32_t magic_z;
    uint32_t magic_mw;     // floor(2^32 / mw), mw = ceil(N / 32): node f <-> (word f % mw, bit f / mw) of the node bitmasks
    // exact 24-bit magic division for the round-3 LDS router (xr_dial3.h): n / d == (n * m24) >> s24 with full-rate 24-bit multiplies,
    // verified at load for every n the kernel can ask for (f < N for Y*Z and mw, r < Y*Z for Z); div24_ok = 0: not found (the host then
    // keeps round 2's form)
    uint32_t m24_yz, m24_z, m24_mw;
    uint32_t s24;          // shift of yz | z << 8 | mw << 16 | div24_ok << 24
    // XR-Maze v2: static guide masks (XrBatchDev::guide_mask, round 5): net n (1-based) of this region at byte gmask_off + (n - 1) * gmask_stride,
    // bit j of byte c = node 8 c + j lies OUTSIDE the net's guide
    int64_t gmask_off;
    int32_t gmask_stride;  // ((N + 7) / 8 rounded up to 16 bytes)
    int32_t pad0;
};

// Workgroup barrier that orders LDS only.  __syncthreads() is a workgroup-scope fence over global memory too: the wave first waits for
// every global store it has in flight (s_waitcnt vmcnt(0)), and under a saturated write stream a store takes ~1.5 us to be acknowledged —
// after every back-trace of a route, between two units of the observation.  Where the barrier only hands LDS data (or nothing but
// control) from wave to wave, this one is enough; global data written before it is NOT visible to the other waves after it.
#if defined(__HIPCC__)
__device__ __forceinline__ void xr_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif

// Packed per-env result of the last step / reset (XR_FETCH_RECORD; layout = xr_step_record of include/xroute_hip.h)
struct XrStepRecord {
    double reward;
    int32_t delta[3];
    int32_t cum[3];
    int32_t nlegal;
    int32_t env_steps;
    int32_t path_len;
    uint8_t done, pad;
    uint16_t status;
};
static_assert(sizeof(XrStepRecord) == 48, "xr_step_record layout");

// Everything a kernel needs, passed by value.
// nets of an env with K nets left whose planes the step kernel writes itself in the split form
#define XR_SPLIT_KEEP(b, K) ((K) - (int)(((int64_t)(K) * (b).obs_split_pm) / 1000))

struct XrBatchDev {
    // regions (static)
    const XrRegionDev* regions;
    const uint32_t* rg_rec;
    int16_t* rg_node_net;    // -1 blockage / 0 normal / net id of the access point
    int16_t* rg_owner0;      // initial owner: 0 free, net id (used AP), XR_OWNER_FOREIGN
    const int32_t* coords;
    const int32_t* net_csr;
    const int32_t* ap_node;
    const int16_t* ap_pin;   // pin + 1
    const int32_t* ap_feat;  // node index | (the access point has an axis neighbour that is an access point of the same net) << 31
    const uint64_t* legal0;
    const uint8_t* net_work; // [like net_csr] predicted route work class of net n of a region (1..255; 0 = no access points), static
    uint8_t* net_meas;       // [like net_csr] MEASURED route work class of net n of a region (0 = never routed yet): written by every route's epilogue
                             // (cycles of the route >> meas_shift, clamped to 1..255), read by the launch orders in place of the geometric guess
                             // net_work — regions replay (10 episodes each, then they come round again), so after one episode every heavy net
                             // is known for what it costs, whatever made it heavy (null: off)
    int32_t meas_shift;
    int32_t heavy_class, heavy_mult;   // LDS router: a net whose measured class is >= heavy_class searches with buckets heavy_mult x as wide (0: off).  Bucket
                                       // widths never change results; a route with many rounds pays per round, one with few pays per re-expansion
    const int32_t* net_info; // [like net_csr] static facts of net n: lowest pin id (pin + 1, 14 bits) | distinct pins << 14 | pins in closed
                             // pockets (never reachable) << 22 | the lowest pin itself is in one << 30   (xr_dial3.h)
    const uint8_t* ap_flags; // [like ap_node] bit 0: the access point's pin sits in a closed pocket (isolated); bits 1..2: which of the three
                             // pin boxes of the search heuristic the access point's pin belongs to (xr_dial3.h)
    const int32_t* guide_csr; // [like net_csr] XR-Maze v2, optional (null: none): boxes of net n are [guide_csr[n], guide_csr[n + 1]) of guide_box
    const int16_t* guide_box; // [boxes][6] x0, y0, x1, y1, z0, z1 (track / layer indices, inclusive)
    const uint8_t* guide_mask; // XR-Maze v2 with guide_cost > 0 (null: none): "outside the guide" of every (region, net) as a bitmask over the nodes, built
                              // once per guide load (xr_guide_mask_kernel) — guide membership is static, so a route reads 1 bit per node with its grid
                              // build instead of testing up to 8 boxes per node in a pass of its own (XrRegionDev::gmask_off)
    int32_t n_regions;
    // envs (mutable)
    int32_t n_envs;
    int32_t n_max;           // owner stride per env (elements), multiple of 8
    int32_t n_lds;           // padded distance-field size (words), max over regions, multiple of 8
    int32_t lw_max;          // words per line bitmask (tracks + columns), max over regions
    int32_t lines_max;       // tracks + columns, max over regions
    int32_t x_max, y_max;    // largest dims over regions (edge-length tables)
    int32_t legal_words;
    int32_t path_cap;
    int32_t* env_region;
    int32_t* env_replay;
    int16_t* owner;
    uint64_t* legal;
    int32_t* nlegal;
    int32_t* cum;            // [B][3]
    int32_t* delta;          // [B][3]
    double* reward;
    uint8_t* done;
    int32_t* status;
    int32_t* path;           // [B][path_cap]
    int32_t* path_len;
    uint64_t* hash;
    int64_t* env_steps;
    unsigned long long* total_steps;
    int32_t* sweeps;
    int32_t* touched;        // [B] nodes whose field word the last route created (HBM-scratch form of the frontier router; else 0)
    XrStepRecord* records;   // [B]
    uint32_t* dist_scratch;  // [B][n_max] only when the distance field does not fit LDS, else null
    uint8_t* cls_scratch;
    unsigned short* list_scratch;   // [B][lines_max] worklists of the large-region variant
    // HBM-scratch form of the frontier router (xr_dial_route_env_big): persistent per-env scratch, CLEAN between routes
    uint32_t* dg_field;      // [B][n_max]
    uint32_t* dg_masks;      // [B][2][n_max/32 + 1]  open bits, cached word minima
    uint32_t* dg_touch;      // [B][n_max]  nodes touched by the current route
    uint32_t* dg_path;       // [B][2][n_max]  path of the current trace; deferred nodes
    long long* phase_cycles; // [B][8] thread-0 cycle counts per kernel phase (only written with -DXR_PHASE_TIMING)
    // fused observation output of the step kernel (null: route only)
    float* obs_out;
    int64_t obs_stride;      // floats per env
    int32_t obs_vec4;        // float4 stores allowed (all N % 4 == 0, aligned planes)
    int32_t obs_head_only;   // 1: split form: the step kernel writes planes 0..1 and the net planes of the lowest ranks,
                             //    xr_netplane_kernel the net planes of the highest floor(K * obs_split_pm / 1000) ranks
    int32_t obs_incremental; // 1: the output buffer holds the previous observation: only planes that change are written (in-place form)
    int32_t obs_split_pm;    // per mille of every env's net planes that go to the writer kernel (1..1000)
    // split observation (xr_plan_kernel -> xr_netplane_kernel): the state every env will have AFTER this step
    int32_t* plan_region;    // [B]
    uint32_t* plan_units;    // [B*k_max] one entry per (env, remaining net): env << 14 | (rank of the net among the remaining)
    int32_t* plan_unit_net;  // [B*k_max] 1-based net id of that unit
    uint32_t* queue;         // [3] next env to route, next unit to write (queue form); number of units (written by xr_plan_kernel)
    int32_t queue_quota_pm;  // units a workgroup writes after each route task, per mille of the average units per env
    int32_t queue_skip_shift; // workgroups with bit `shift` of their index set start with units instead of a route (-1: none)
    int32_t obs_lds_bytes;   // dynamic LDS of the launch that runs xr_obs_epilogue (0: unknown -> the epilogue reads the state rows from global memory)
    int32_t queue_grid;      // workgroups of the queue-form step kernel of this call (their first tasks are static: xr_queue_first)
    // parameters
    int32_t via_cost, pen_cost, max_route_count, auto_reset;
    const int32_t* route_order;   // route kernel: workgroup i routes env route_order[i] (null: env_base + i); longest predicted first
    int32_t env_base, env_count;   // route kernel: envs [env_base, env_base + env_count) (env_count 0 = all); stream-per-region mode
    int32_t guide_cost, guide_margin, maze_end_iter;   // XR-Maze v2 knobs (0, 0, 1 = XR-Maze v1)
    int32_t dial_mult_big;   // the same for the HBM-scratch form
    int32_t dial_mult;       // bucket width of the frontier router in units of the region's smallest edge length
    int32_t round_cap;       // relaxation rounds one search may take before the router aborts (0: 1024 + N), XR_ENV_ROUTER_ABORT
    // window form of the LDS router for regions that do not fit LDS (xr_dial3.h, WIN; 0 = off): window of win_x x win_y tracks on every
    // layer, win_nmax = its nodes padded to 8, margin kept around the net's access-point box, rows start at multiples of win_ystep
    int32_t win_x, win_y, win_nmax, win_margin, win_ystep;
    uint32_t win_m24_yz, win_m24_z, win_m24_mw, win_s24;   // exact 24-bit magics of the window's Y*Z, of Z and of its mask-word count (shifts packed like XrRegionDev::s24)
    double w_violation, w_via, w_wirelength;
};
