/*
 * xroute_hip.h — C ABI of libxroute_hip.so: the MI355X (gfx950) in-process replacement for the
 * reference's env hot path  Game.reset()/Game.step()  (simulator round trip + build_3Dgrid + metric
 * deltas + reward).
 *
 * The reference has no FFI for this path: the seam is the Python `Game` object and the protobuf
 * `Message` behind it.  Every entry point below names the reference interface it
 * replaces.  All paths are relative to the reference repository root.
 *
 * Conventions
 *   - plain C types only; every function returns an int32_t status (XR_OK == 0, negative = error);
 *     xr_last_error() returns a thread-local message for the last failing call on this thread;
 *   - no exception crosses the ABI;
 *   - `stream` arguments are a hipStream_t passed as void* (pass torch's current stream);
 *     functions only enqueue work on it and never synchronise unless documented ("sync");
 *   - pointers named *_dev are device pointers on the batch's device, *_host are host pointers;
 *     the caller owns every buffer it passes; the library owns batch state until xr_batch_destroy;
 *   - calls on one batch must be serialised by the caller; one batch is bound to one HIP device;
 *   - net ids are 1-based at this boundary exactly as in the reference's env API
 *     (baseline/baseline_utils.py:20,26,33 shift the wire's 0-based ids by +1; Game.step sends
 *     action-1, :410).
 *
 * Node records (uint32), flat node order f = (x*Y + y)*Z + z  (the memory order of the reference
 * observation: baseline/build_3Dgrid.py:97-103 reshapes a zeros([X,Y,Z]) tensor without permuting):
 *   bits  1:0  type  0 BLOCKAGE 1 NORMAL 2 ACCESS     (baseline/openroad_api/proto/net_ordering.proto:5-9)
 *   bit   2    is_used                                (net_ordering.proto:24)
 *   bits 16:3  net+1 (0 = none)                       (net_ordering.proto:25)
 *   bits 30:17 pin+1 (0 = none)                       (net_ordering.proto:26)
 */
#ifndef XROUTE_HIP_H
#define XROUTE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XR_ABI_VERSION 9

/* status codes */
#define XR_OK            0
#define XR_ERR_INVALID  (-1)   /* bad argument */
#define XR_ERR_NOMEM    (-2)   /* host or device allocation failed */
#define XR_ERR_HIP      (-3)   /* a HIP runtime call failed (message has the HIP error string) */
#define XR_ERR_STATE    (-4)   /* call order violated (e.g. step before load_regions) */
#define XR_ERR_RANGE    (-5)   /* size/limit exceeded (dims, net count, LDS capacity ...) */
#define XR_ERR_PARSE    (-6)   /* malformed protobuf bytes */

/* node record fields */
#define XR_TYPE_BLOCKAGE 0u
#define XR_TYPE_NORMAL   1u
#define XR_TYPE_ACCESS   2u
#define XR_REC_TYPE(r)   ((r) & 3u)
#define XR_REC_USED(r)   (((r) >> 2) & 1u)
#define XR_REC_NET1(r)   (((r) >> 3) & 0x3FFFu)    /* net + 1, 0 = none */
#define XR_REC_PIN1(r)   (((r) >> 17) & 0x3FFFu)   /* pin + 1, 0 = none */
#define XR_MAX_NETS      16382
#define XR_MAX_LAYERS    32

/* owner value of nodes occupied by something that is not a net of this region
 * (blockages, pre-routed wires: is_used NORMAL nodes in the initial Request) */
#define XR_OWNER_FOREIGN 0x7FFF

/* per-env status bits (XR_FETCH_STATUS) */
#define XR_ENV_OK            0
#define XR_ENV_BAD_ACTION    1   /* action not in the legal set: step was a no-op (the reference does
                                    not check this client-side, baseline/baseline_utils.py:409-412) */
#define XR_ENV_UNREACHABLE   2   /* at least one pin could not be reached */
#define XR_ENV_PATH_TRUNC    4   /* path longer than path_cap: path list truncated, metrics exact */
#define XR_ENV_WAS_RESET     8   /* auto_reset: this step re-initialised the env instead of routing */
#define XR_ENV_ROUTER_ABORT 16   /* the router gave up on this net: a search exceeded its round cap (1024 + N relaxation rounds, far
                                    beyond anything a legal region needs; xr_config.debug_round_cap forces it).  The pins not
                                    connected yet are charged as unreachable; the env stays consistent and the device never
                                    spins on a pathological region */

typedef struct xr_batch xr_batch;

/* Router / env parameters.  The simulator knobs mirror the reference's TCL
 * (ispd/ispd18_test1/run-net-ordering-training.tcl:3 `-drc_cost 8`), the reward weights its
 * trainers (baseline/DQN/train_DQN.py:98-99, baseline/PPO/train_PPO.py:101-102), max_route_count
 * its control plane (examples/launch_training.py:28). */
#define XR_ROUTER_SWEEP 1
#define XR_ROUTER_DIAL  2
#define XR_ROUTER_DIAL_R2 3   /* round 2's LDS form of the frontier router (four waves per search, mask scans): kept for A/B runs */
#define XR_OBS_FUSED 1
#define XR_OBS_SPLIT 2
#define XR_OBS_QUEUE 3
typedef struct xr_config {
    int32_t struct_size;      /* = sizeof(xr_config); checked */
    int32_t device;           /* HIP device ordinal */
    int32_t n_envs;           /* B: env slots in this batch */
    int32_t via_cost;         /* XR-Maze v1: cost of one via edge, DBU-equivalent (default 800) */
    int32_t drc_cost;         /* default 8 */
    int32_t drc_unit;         /* DBU per drc_cost unit (default 400): entering a node held by
                                 another net costs drc_cost*drc_unit and counts one violation */
    int32_t max_route_count;  /* replays of one region before rotating to the next (default 10) */
    int32_t auto_reset;       /* 1: xr_batch_step re-initialises envs that were done (vector env) */
    int32_t path_cap;         /* max recorded path nodes per env-step (0 = min(N, 4096)) */
    int32_t block_threads;    /* route kernel workgroup size, 0 = default */
    int32_t force_scratch_field; /* 1: keep the distance field in HBM scratch even when it would fit LDS (the
                                    large-region code path; for tests and A/B measurements) */
    int32_t obs_mode;         /* xr_batch_step_observe: 0 = default (XR_OBS_QUEUE where it applies, the fastest as measured,
                                 else XR_OBS_FUSED), XR_OBS_FUSED = one launch of one workgroup per env, XR_OBS_SPLIT = route
                                 kernel + concurrent net-plane writer, XR_OBS_QUEUE = planning kernel + one persistent launch */
    double  w_violation;      /* 500  */
    double  w_via;            /* 4    */
    double  w_wirelength;     /* 0.5  */
    int32_t obs_writer_blocks; /* XR_OBS_SPLIT: workgroups of the net-plane writer; XR_OBS_QUEUE: workgroups of the
                                  persistent launch (0 = default: 512 / as many as are resident on the chip) */
    int32_t router;           /* XR-Maze v1 relaxation scheme (same results, bit for bit): 0 = auto (XR_ROUTER_DIAL wherever it applies;
                                 only the full-rewrite queue launch of a batch of >= 4096 slots takes XR_ROUTER_SWEEP, measured
                                 1-2 % faster there), XR_ROUTER_SWEEP = line-segment sweeps over dirty-line
                                 worklists (round 1), XR_ROUTER_DIAL = bucketed frontier expansion (Dial's algorithm with A* keys; regions that fit
                                 LDS take round 3's form — mask rounds, quads of lanes per chain, predecessor directions in the field word — whenever one
                                 edge with its penalties stays below 2^20; larger regions the HBM-scratch form), XR_ROUTER_DIAL_R2 = round 2's LDS form of it */
    int32_t dial_mult;        /* XR_ROUTER_DIAL: bucket width in units of the region's smallest edge length (0 = default 8) */
    int32_t guide_cost;       /* XR-Maze v2 (all three neutral by default = XR-Maze v1): entering a node outside the net's guide costs this
                                 much extra, DBU-equivalent (the role of `-follow_guide 1`, run-net-ordering-training.tcl:3); the guide of
                                 a net = bounding box of all its access points in track indices, inflated by guide_margin, every layer */
    int32_t guide_margin;
    int32_t maze_end_iter;    /* >= 1 (`-maze_end_iter 3 -ripup_mode 1` of the same line): attempt t routes the net with the penalty
                                 drc_cost*drc_unit << t; an attempt whose path uses a node held by another net is ripped up unless it is
                                 the last one.  Frontier router only (XR_ROUTER_SWEEP: XR_ERR_RANGE) */
    int32_t stream_per_region; /* 1: "one region per stream" (north_star's first partition): xr_batch_step / _step_observe (fused
                                  form) / _step_compact launch ONE single-workgroup kernel per env slot, round-robin over a pool of
                                  internal HIP streams, joined to the caller's stream by events.  For batches of <= 64 slots only
                                  (XR_ERR_RANGE above); measured against the default one-launch form in DESIGN.md */
    int32_t obs_helper_blocks; /* XR_OBS_QUEUE: LDS-free helper-writer workgroups launched beside the persistent step kernel on an
                                  internal stream, draining the same unit queue (0 = none, the default: measured no faster) */
    int32_t obs_split_permille; /* XR_OBS_SPLIT: per mille of every env's net planes (its highest-ranked nets) that the
                                   writer kernel streams; the step kernel writes the rest after routing (0 = 1000 = all).
                                   XR_OBS_QUEUE: units a workgroup writes after each route task, per mille of the average
                                   number of units per env (0 = 750) */
    int32_t launch_order;     /* route-only launches (xr_batch_step, xr_batch_step_compact, fused xr_batch_step_observe): order in
                                 which the env slots are handed to workgroups.  0 = default: longest predicted route first when
                                 the batch has more slots than the chip holds workgroups (a ~10 us ordering kernel ahead of the
                                 launch; the prediction is the chosen net's bounding box and pin count), slot order otherwise;
                                 1 = slot order always; 2 = longest first always.  The queue form of xr_batch_step_observe hands its
                                 ROUTE TASKS out the same way: 0 = longest first for batches of at most 2 routes per resident workgroup (4
                                 with planes that are not 16-byte aligned), slot order above that; 1 / 2 as above.  Results do not depend on it.  (Takes the
                                 struct's former tail padding: sizeof(xr_config) is unchanged.) */
    int32_t debug_round_cap;  /* 0 = default.  > 0: relaxation rounds one search of the router may take before it aborts with
                                 XR_ENV_ROUTER_ABORT (tests force the abort path with 1) */
    int32_t window;           /* regions whose distance field does not fit LDS (BASELINE config 5), > 0: the router first runs inside an LDS window
                                 of the region centred on the net (the largest square window of at most this many tracks that fits LDS: 52 at 12
                                 layers) and accepts the result only with an exactness certificate — no shortest path to anything the step looks
                                 at leaves the window — else the HBM-scratch form routes the net.  0 (default) / < 0: off — built and bit-exact,
                                 but measured no faster than the HBM-scratch form alone on config 5.  Results never depend on
                                 it.  (ABI 6; was reserved0 = 0.) */
} xr_config;

/* One region = one simulator Request (net_ordering.proto:29-45) in dense form; host pointers. */
typedef struct xr_region_desc {
    int32_t dim_x, dim_y, dim_z;
    const int32_t*  xs_host;        /* [dim_x] track coordinates (Node.point_x), strictly increasing */
    const int32_t*  ys_host;        /* [dim_y] */
    const uint8_t*  layer_dir_host; /* [dim_z] 0 = horizontal (x moves), 1 = vertical (y moves) */
    const uint32_t* nodes_host;     /* [X*Y*Z] packed records */
    int32_t n_nets;                 /* nets 1..n_nets may appear in records */
    int32_t metrics0[3];            /* cumulative (violation, wirelength, via) of the initial Request
                                       (Request.reward_*, net_ordering.proto:36-38) */
} xr_region_desc;

/* what xr_batch_fetch copies (device -> caller's buffer: a DEVICE buffer or PINNED host memory; async on `stream`) */
#define XR_FETCH_CUM       0   /* int32 [B][3]  cumulative (violation, wirelength, via)  = data[2] */
#define XR_FETCH_DELTA     1   /* int32 [B][3]  last step's deltas  (Game.step :426-433) */
#define XR_FETCH_REWARD    2   /* double[B]     -(wv*dvio + wvia*dvia + wwl*dwl)  (train_DQN.py:98-99) */
#define XR_FETCH_DONE      3   /* uint8 [B]     len(netSet)==0  (Game.step :435-436) */
#define XR_FETCH_NLEGAL    4   /* int32 [B]     len(netSet) */
#define XR_FETCH_STATUS    5   /* int32 [B]     XR_ENV_* bits of the last step */
#define XR_FETCH_LEGAL     6   /* uint64[B][legal_words] bit n-1 set <=> net n in netSet */
#define XR_FETCH_PATH_LEN  7   /* int32 [B]     nodes claimed by the last step (untruncated count) */
#define XR_FETCH_PATH      8   /* int32 [B][path_cap] flat node ids in back-trace order */
#define XR_FETCH_OWNER     9   /* int16 [B][n_max] per-node owner (0 free, net id, XR_OWNER_FOREIGN) */
#define XR_FETCH_HASH     10   /* uint64[B]     FNV-1a chain over every (path, deltas) since reset_all */
#define XR_FETCH_REGION   11   /* int32 [B]     region index the env currently plays */
#define XR_FETCH_STEPS    12   /* int64 [1]     env-steps (real routes, not resets) since create */
#define XR_FETCH_SWEEPS   13   /* int32 [B]     relaxation sweeps used by the last step */
#define XR_FETCH_RECORD   15   /* xr_step_record[B]  everything a caller needs after a step or a reset, one 48-byte
                                                 record per env (written by the kernels themselves: one copy, one sync) */
#define XR_FETCH_TOUCHED  16   /* int32 [B]     nodes whose field word the last route created (HBM-scratch form of the frontier
                                                 router: the work it really did; 0 for the other forms) */
#define XR_FETCH_UNITS    17   /* uint32[1]     net-plane units (7 planes of one net of one env) the last xr_batch_step_observe* planned */
#define XR_FETCH_ROUTE_ORDER 18 /* int32[B]     env slots in the order the last longest-first route-only launch handed them out (xr_config.launch_order) */
#define XR_FETCH_REPLAY   19   /* int32 [B]     replays of the current region so far (region rotation, examples/launch_training.py:28-46) */
#define XR_FETCH_ENV_STEPS 20  /* int64 [B]     real steps of every env slot since create */
#define XR_FETCH_PHASES   14   /* int64 [B][8]  slot 7: shader-clock cycles of the env's last route (frontier router; what the measured launch order
                                                 is built from — bench.py's launch utilisation = mean / max over a launch); slots 0..6: route-kernel
                                                 phase cycle counts, zero unless the library was built with -DXR_PHASE_TIMING (which also owns slot 7) */

/* Per-env result of the last xr_batch_step* / xr_batch_reset: what Game.step returns besides the observation
 * (baseline/baseline_utils.py:426-439) plus what Game.reset records (:472-473), packed for one transfer — also the
 * record that the multi-GPU gather exchanges (one all_gather of B x 48 bytes per step). */
typedef struct xr_step_record {
    double   reward;       /* -(500 dvio + 4 dvia + 0.5 dwl), train_DQN.py:98-99 */
    int32_t  delta[3];     /* violation, wirelength, via of the last step */
    int32_t  cum[3];       /* cumulative metrics (data[2]) */
    int32_t  nlegal;       /* len(netSet) */
    int32_t  env_steps;    /* real steps of this env slot since create (low 32 bits) */
    int32_t  path_len;
    uint8_t  done;
    uint8_t  pad;
    uint16_t status;       /* XR_ENV_* bits */
} xr_step_record;

int32_t     xr_abi_version(void);
const char* xr_last_error(void);
void        xr_config_default(xr_config* cfg);
int32_t     xr_device_count(int32_t* n);

/* ---- batch life cycle ------------------------------------------------------------------- */
int32_t xr_batch_create(const xr_config* cfg, xr_batch** out);
int32_t xr_batch_destroy(xr_batch* b);

/* Replaces: receiving the initial Request of each region dump (examples/launch_training.py:57-64
 * relaunching the simulator on a `workerx*_y*` dir; baseline/baseline_utils.py:459-466).
 * Uploads the regions and runs the ingest kernel (records -> compact node_net/owner0 state + per-net
 * access-point lists).  Env e initially plays region e % n_regions.  Synchronises `stream`. */
int32_t xr_batch_load_regions(xr_batch* b, const xr_region_desc* regions_host, int32_t n_regions,
                              void* stream);

/* env -> region assignment (host array of n_envs region indices); takes effect at the next reset */
int32_t xr_batch_assign(xr_batch* b, const int32_t* env_region_host);

/* Sizes the caller needs for its buffers. */
int32_t xr_batch_sizes(const xr_batch* b, int32_t* n_envs, int32_t* n_regions, int32_t* n_max,
                       int32_t* k_max, int32_t* legal_words, int32_t* path_cap, int64_t* obs_env_stride);

/* Replaces Game.reset (baseline/baseline_utils.py:441-481) + the control plane's region rotation
 * (examples/launch_training.py:33-54).  mask_dev: uint8[B] (nonzero = reset this env) or NULL = all.
 * rotate != 0 applies the 10-replays-then-next-region policy, 0 replays the assigned region. */
int32_t xr_batch_reset(xr_batch* b, const uint8_t* mask_dev, int32_t rotate, void* stream);

/* Replaces Game.step's simulator round trip (baseline/baseline_utils.py:409-419: send net_index =
 * action-1, receive the next Request) and its metric bookkeeping (:426-438): routes net
 * actions_dev[e] (1-based) in env e with the XR-Maze v1 router, claims the path nodes, accumulates
 * wirelength/via/violation, updates netSet/done/reward. */
int32_t xr_batch_step(xr_batch* b, const int32_t* actions_dev, void* stream);

/* Game.step with its observation (baseline/baseline_utils.py:409-423: route, then build_3Dgrid on the new state),
 * out_dev / env_stride as in xr_batch_observation, for envs [0, n_envs).  Three forms, same bytes:
 *   XR_OBS_FUSED  one launch: every workgroup routes its env, then streams that env's observation.
 *   XR_OBS_SPLIT  the 7K net planes of an env do not depend on the routing result (they are functions of the
 *                 region's static node array and of which nets remain, which follows from the state before the
 *                 step), so a planning kernel derives every env's post-step net set, a balanced, address-ordered
 *                 writer kernel streams all net planes on an internal stream, and the route kernel — running
 *                 concurrently on the caller's stream — writes planes 0..1.  The caller's stream is joined with
 *                 the internal one before the call returns control of the stream (event wait, no host sync).
 *                 Needs a 16-byte aligned out_dev and env_stride % 4 == 0 (regions whose N is not a multiple of 4 take
 *                 a writer that resolves plane boundaries per float); otherwise the call runs XR_OBS_FUSED.
 *   XR_OBS_QUEUE  (default) the same planning kernel, then ONE persistent launch on the caller's stream whose workgroups
 *                 drain two task queues: envs to route (+ their planes 0..1) and net-plane units to write.  Units do not
 *                 depend on routing, so they keep HBM busy while other workgroups route and the launch ends in a
 *                 fine-grained pure-write drain instead of a tail of whole envs.  Same requirements as XR_OBS_SPLIT. */
int32_t xr_batch_step_observe(xr_batch* b, const int32_t* actions_dev, float* out_dev, int64_t env_stride,
                              void* stream);

/* In-place form of xr_batch_step_observe for a caller that keeps ONE observation buffer for the batch (the vector env): the
 * planes of a net are static (baseline/build_3Dgrid.py:106-142) and the channel order is "nets ascending" (:177-179), so removing
 * the routed net shifts only the nets ABOVE it down by one 7-plane slot — the nets below keep their slot and their bytes.
 * When out_dev / env_stride are the buffer that received this batch's previous full observation (the last xr_batch_observation
 * over all slots, xr_batch_step_observe or xr_batch_step_observe_inplace, with no other state-changing call in between) and
 * the caller has not written to it, only planes 0..1 and the planes of the remaining nets above the routed one are written
 * (a slot that re-initialises writes everything, a rejected action planes 0..1 only): the buffer ends up byte-identical to
 * what xr_batch_step_observe writes, with about half the HBM traffic under a uniform net choice.  Any other buffer: a full
 * write, exactly xr_batch_step_observe.  xr_batch_observe_timing reports mode | 16 when the in-place path ran (and mode | 32
 * when the auto router ran the line-segment sweeps in the queue launch: full rewrite of a batch of >= 4096 slots). */
int32_t xr_batch_step_observe_inplace(xr_batch* b, const int32_t* actions_dev, float* out_dev, int64_t env_stride,
                                      void* stream);

/* Compact-consumer mode.  The reference consumer re-encodes every net's 7 planes at every step (baseline/DQN/DQN.py:138-155),
 * but those planes are functions of the region's static access points only (baseline/build_3Dgrid.py:106-142): a consumer
 * that caches per-(region, net) results (xroute_env_amd.agents.NetVectorCache) needs, per step, only the two planes that
 * change — plane 0 (obstacles) and plane 1 (net order).
 *   xr_batch_step_compact  = xr_batch_step + planes 0..1 of every env at head_out_dev + e*head_stride (floats; 16-byte
 *                            aligned buffer, head_stride % 4 == 0, head_stride >= 2*n_max): 8·N bytes per env-step instead
 *                            of 4·N·(2+7K).
 *   xr_batch_net_planes    = the 7 planes [7, Z, Y, X] of n_pairs (region index, 1-based net id) pairs, pair i at
 *                            out_dev + i*pair_stride (floats, pair_stride >= 7*n_max) — byte-identical to planes 2+7i..8+7i
 *                            of xr_batch_observation for an env playing that region with that net at rank i. */
int32_t xr_batch_step_compact(xr_batch* b, const int32_t* actions_dev, float* head_out_dev, int64_t head_stride,
                              void* stream);
int32_t xr_batch_net_planes(xr_batch* b, const int32_t* pair_region_dev, const int32_t* pair_net_dev, int32_t n_pairs,
                            float* out_dev, int64_t pair_stride, void* stream);

/* Compact state for a CENTRAL learner (ABI 7; SURVEY.md §8e, BASELINE config 4).  The reference's caller loop evaluates its policy on the
 * observation of the env it just stepped (baseline/PPO/train_PPO.py:96-99: `action = ppo_agent.select_action(state); state, ... = game.step(action)`).
 * When ONE learner process evaluates the policy for env slots that live on other GPUs, what has to travel per env-step is not the two fp32
 * planes that change (8·N bytes) but what they are functions of: plane 0 (obstacle: blockage or used, baseline/build_3Dgrid.py:19-36,94-103) is
 * one bit per node, plane 1 (the remaining nets' ids ascending at flat positions 0..K-1, :144-161) is the legal-net bitmask.
 *   xr_batch_state_row_bytes  bytes of one packed row of THIS batch: 16 + 8·(legal_words + ceil(n_max / 64)); ranks agree on the maximum.
 *   xr_batch_pack_state       one row per env slot at rows_dev + e*row_bytes (8-byte aligned, row_bytes % 8 == 0):
 *                               int32 region + region_base | int32 nlegal | int32 legal_words | int32 occ_words |
 *                               uint64 legal[legal_words] | uint64 occ[occ_words]  (bit f%64 of word f/64 = node f, flat observation order)
 *                             region_base: what turns this batch's local region index into an index of the learner's region table.
 *   xr_batch_expand_state     (learner side; `b` supplies the region table: it must hold every region the rows name) n_rows rows -> the
 *                             fp32 head rows xr_batch_step_compact writes for those envs, byte for byte (planes 0..1 at
 *                             head_out_dev + i*head_stride, each env in its own region's layout), plus nlegal / region of every row as
 *                             int32 arrays for xr_agent_actor.  A row that does not parse (region outside the table, sizes that do not
 *                             fit, nlegal != popcount(legal), a legal bit beyond the region's nets) is left unwritten and flagged
 *                             nlegal = region = -1. */
int32_t xr_batch_state_row_bytes(const xr_batch* b, int64_t* row_bytes);
/* The client half of Game.step WITHOUT the route (ABI 9; BASELINE config 2: "grid-build + reward only"): a new state of every env slot, as an
 * external simulator's `Request` carries it, becomes the batch's state — replaces baseline/baseline_utils.py:420-438 for a batch (`data =
 * handle_messange(..)`; the metric deltas `*_cur_step - *_last_step`, :426-433; `done = len(netSet) == 0`, :435-436) with the trainers' reward
 * (baseline/DQN/train_DQN.py:98-99, double) computed in the same kernel.  Follow it with xr_batch_observation for the fp32 grid (build_3Dgrid, :423).
 *   owner_dev   int16 [n_envs][n_max]        occupancy of every node in flat order: 0 free, the 1-based net id holding it, XR_OWNER_FOREIGN
 *                                            (16-byte aligned rows: n_max is a multiple of 8)
 *   legal_dev   uint64 [n_envs][legal_words] bit n-1 = net n is still to route (`Request.nets`, +1); bits beyond the region's nets are dropped
 *   cum_dev     int32 [n_envs][3]            cumulative violation / wirelength / via (`reward_violation`, `reward_wire_length`, `reward_via`)
 * Afterwards XR_FETCH_RECORD / _DELTA / _REWARD / _DONE / _NLEGAL describe the step; env_steps advanced by one; the hash chain (routed paths) is
 * untouched.  Enqueues one kernel, never synchronises. */
int32_t xr_batch_ingest_state(xr_batch* b, const int16_t* owner_dev, const uint64_t* legal_dev, const int32_t* cum_dev, void* stream);
int32_t xr_batch_pack_state(xr_batch* b, uint8_t* rows_dev, int64_t row_bytes, int32_t region_base, void* stream);
int32_t xr_batch_expand_state(xr_batch* b, const uint8_t* rows_dev, int64_t row_bytes, int32_t n_rows, float* head_out_dev, int64_t head_stride,
                              int32_t* nlegal_out_dev, int32_t* region_out_dev, void* stream);

/* Whole-order re-route, the step of the reference's two other env contracts: the A3C env answers the simulator with
 * a complete net list (baseline/A3C/utils.py:305-307, Response.net_list of net_ordering.proto v2 field 2) and the
 * MCTS dispatcher re-routes the region from scratch with `routed_nets + unrouted_nets` after every selection
 * (baseline/xroute/trainer4/dispatcher.py:113-118).  Every env restores its assigned region's initial state and
 * routes orders_dev[e*stride + 0..] (int32, 1-based net ids, a value <= 0 ends the list; stride >= k_max) in that
 * order, in one launch.  Afterwards XR_FETCH_CUM holds the final cumulative metrics, XR_FETCH_DELTA / REWARD the
 * totals over the whole order (cum - initial), XR_FETCH_STATUS the OR over the routed nets, XR_FETCH_PATH_LEN /
 * SWEEPS the sums, XR_FETCH_DONE whether every net was routed.  Illegal entries (out of range, repeated) are skipped and
 * flagged XR_ENV_BAD_ACTION.  net_stats_dev: NULL or int32[n_envs][stride][4], row = net id - 1:
 * {d_violation, d_wirelength, d_via} of that net's route in this order (the simulator's `metrics_delta`, proto v2
 * field 13) and a counter incremented once per route (`count_map`, field 12); rows of nets not routed are left
 * untouched, the caller zeroes the buffer when an episode starts. */
int32_t xr_batch_route_order(xr_batch* b, const int32_t* orders_dev, int32_t stride, int32_t* net_stats_dev,
                             void* stream);

/* Placement of the step kernel for the loaded regions: resident workgroups per CU as the HIP runtime computes it
 * (hipOccupancyMaxActiveBlocksPerMultiprocessor) and LDS bytes per workgroup (dynamic + static).  ispd18_test1-sized
 * regions are laid out for 4 workgroups per CU (4 x 39.9 KB of the 160 KB); a build that loses that is slower by a
 * quarter, so the tests check it. */
int32_t xr_batch_route_occupancy(xr_batch* b, int32_t* workgroups_per_cu, int64_t* lds_bytes_per_workgroup);

/* Duration in milliseconds (HIP events on the library's internal stream) of the net-plane writer kernel that the
 * last xr_batch_step_observe launched in XR_OBS_SPLIT mode; blocks until that kernel has finished.  *mode_out = the
 * mode that call ran in (XR_OBS_FUSED / XR_OBS_SPLIT); *writer_ms = 0 for XR_OBS_FUSED. */
int32_t xr_batch_observe_timing(xr_batch* b, int32_t* mode_out, float* writer_ms);

/* BASELINE config "random net-order policy": actions_dev[e] = a uniformly chosen legal net of env e
 * (1-based; 0 when the env is done), from a counter-based hash of (seed, e, step count). */
int32_t xr_batch_random_actions(xr_batch* b, int32_t* actions_dev, uint64_t seed, void* stream);

/* Replaces build_3Dgrid on the current state (baseline/build_3Dgrid.py:224-270, called at
 * baseline/baseline_utils.py:422-423,469-470): fp32 observation [2+7K, Z, Y, X] of envs
 * [env_lo, env_hi) written at out_dev + (e-env_lo)*env_stride (floats); K = nlegal[e]. */
int32_t xr_batch_observation(xr_batch* b, float* out_dev, int64_t env_stride, int32_t env_lo,
                             int32_t env_hi, void* stream);

int32_t xr_batch_fetch(xr_batch* b, int32_t what, void* dst_dev, size_t dst_bytes, void* stream);

/* Env-state restore: the inverse of xr_batch_fetch for the arrays that ARE the state of the env slots — XR_FETCH_OWNER, _LEGAL,
 * _NLEGAL, _CUM, _DELTA, _REWARD, _DONE, _STATUS, _PATH_LEN, _HASH, _REGION, _REPLAY, _ENV_STEPS, _RECORD (any other selector:
 * XR_ERR_INVALID).  src_dev: a DEVICE buffer or pinned host memory of exactly the array's size; async on `stream`.  The
 * reference never checkpoints its env (the state lives in the simulator process; its agents do: baseline/DQN/DQN.py:236-242):
 * here the whole batch state is a handful of device arrays, so a dump (fetch) / restore (store) pair is the env checkpoint —
 * a batch restored into a twin created with the same config and regions continues bit-identically (hash chains included). */
int32_t xr_batch_store(xr_batch* b, int32_t what, const void* src_dev, size_t src_bytes, void* stream);

/* ---- XR-Maze v2: global-route guides (optional) ---------------------------------------------- */
/* The reference's simulator runs with `-follow_guide 1` (ispd/ispd18_test1/run-net-ordering-training.tcl:3) on the guide file
 * it ships (ispd/ispd18_test1/ispd18_test1.input.guide: per net, rectangles per metal layer).  With xr_config.guide_cost > 0 a
 * net's guide is, by default, the bounding box of its access points; this call replaces it, per (region, net), by up to
 * XR_GUIDE_MAX_BOXES boxes (x0, y0, x1, y1, z0, z1: track / layer indices of the region's grid, inclusive — the guide
 * rectangles clipped to the region, xroute_env_amd/lefdef.py): a node is inside the guide when it lies in any box inflated by
 * guide_margin tracks in x and y.  box_off_host[r] = int32[n_nets(r) + 1] offsets (0-based nets) into boxes_host[r] =
 * int16[boxes][6]; box_off_host[r] == NULL: region r keeps the default guides; box_off_host == NULL: drop every guide.
 * A net with no box keeps the default guide.  Call after xr_batch_load_regions (a reload drops the guides); host pointers,
 * copied before the call returns.  Build-defined like the rest of XR-Maze (parity unpinned against the reference's router). */
#define XR_GUIDE_MAX_BOXES 8
int32_t xr_batch_load_guides(xr_batch* b, const int32_t* const* box_off_host, const int16_t* const* boxes_host, void* stream);

/* ---- stateless observation build --------------------------------------------------------- */
/* Replaces build_3Dgrid(data, routed_nets, bool_inference) for a caller that already holds the
 * node records (the inference servers baseline/DQN/test_DQN.py:54-62, baseline/PPO/test_PPO.py:
 * 58-62): nets_dev = the K legal 1-based net ids in ascending order. out_dev: (2+7K)*N floats. */
int32_t xr_observation_from_records(const uint32_t* nodes_dev, int32_t dim_x, int32_t dim_y,
                                    int32_t dim_z, const int32_t* nets_dev, int32_t k,
                                    float* out_dev, void* stream);

/* ---- agent side (a consumer of the path) ----------------------------------- */
/* The obstacle tower of the reference's RepresentationNetwork (baseline/baseline_utils.py:231-379: ob_conv1 -> ob_align_conv1 ->
 * ob_conv2 -> ob_align_conv2, what its DQN / PPO agents run on plane 0 of every observation before choosing a net) as one fused
 * kernel, eval mode: head_dev fp32 [n_envs][head_stride] with plane 0 of every env first (the buffers of xr_batch_step_compact /
 * xr_batch_step_observe), (D, H, W) = the observation tensor's trailing dims (dim_z, dim_y, dim_x), weights_dev =
 * xr_agent_obstacle_tower_weights() floats packed by xroute_env_amd/agents.py FusedObstacleTower.refresh (BatchNorm folded; the two 7 -> 7-channel
 * convolutions as the per-lane A operands of the kernel's v_mfma_f32_16x16x4_f32 steps: csrc/xr_agent.hip XT_* offsets), out_dev fp32 [n_envs][64]
 * (normalize != 0: after the reference's row-wise min-max normalisation, baseline/baseline_utils.py:45-63).  XR_ERR_RANGE: a grid shape the kernel does not take (use the framework path). */
int32_t xr_agent_obstacle_tower_weights(void);
int32_t xr_agent_obstacle_tower(const float* head_dev, int64_t head_stride, int32_t n_envs, int32_t dim_d, int32_t dim_h, int32_t dim_w,
                                const float* weights_dev, float* out_dev, int32_t normalize, void* stream);
/* The actor head (baseline/DQN/DQN.py:27-46 `Actor`: mlp 128 -> 128 -> 64 -> 1, ELU, on [state vector ++ net vector]) for every legal net of
 * every env + the greedy action (DQN.inference_action; PPO samples from the same logits): state_dev fp32 [n_envs][64] normalised state
 * vectors, the net ids from the net-order channel of head_dev (plane 1, `ids_off` floats into an env's row), the normalised net vectors from
 * cache_vec_dev [regions * cache_kmax][64] (row = region * cache_kmax + net - 1: agents.NetVectorCache, complete) or, when cache_pre_dev
 * [regions * cache_kmax][128] is given, from the first layer's net half already applied to them (W1[:, 64:] . vec, once per weight update), weights_dev =
 * xr_agent_actor_weights() floats (transposed by agents.FusedActorHead).  logits_dev: optional fp32 [n_envs][kcap], -inf beyond an env's
 * nets; action_dev int32 [n_envs]: first maximum in the channel's order, 0 without nets. */
int32_t xr_agent_actor_weights(void);
int32_t xr_agent_actor(const float* state_dev, const float* head_dev, int64_t head_stride, int32_t ids_off, const int32_t* nlegal_dev,
                       const int32_t* region_dev, const float* cache_vec_dev, const float* cache_pre_dev, int32_t cache_kmax, const float* weights_dev,
                       int32_t n_envs, int32_t kcap, float* logits_dev, int32_t* action_dev, void* stream);
/* ABI 8: the same kernel with PPO's rollout sampling inside (baseline/PPO/PPO.py:124-146 `ActorCritic.act`: `dist = Categorical(action_probs); action =
 * dist.sample()`; caller `select_action`, :205-217).  env_ids_dev int64 [n_envs] = the GLOBAL id of every env, s0 = the (seed, step) word of
 * xroute_env_amd.agents.counter_uniform: action_dev = a sample of Categorical(softmax(logits)) by the Gumbel-max trick over counter-based uniforms of
 * (s0, global env id, rank of the net) — no generator state, so an env gets the same action whichever rank or batch evaluates it, and the same one the
 * framework path's tensor ops choose (same integer hash, same float operations in the same order).  env_ids_dev == NULL: the greedy action. */
int32_t xr_agent_actor_sample(const float* state_dev, const float* head_dev, int64_t head_stride, int32_t ids_off, const int32_t* nlegal_dev,
                              const int32_t* region_dev, const float* cache_vec_dev, const float* cache_pre_dev, int32_t cache_kmax, const float* weights_dev,
                              int32_t n_envs, int32_t kcap, float* logits_dev, int32_t* action_dev, const int64_t* env_ids_dev, uint64_t s0, void* stream);

/* The NET tower of the same network (ABI 9: `net_conv1 -> net_align_conv1 -> net_conv2 -> net_align_conv2`, baseline/baseline_utils.py:246-262, 350-357) for
 * n_pairs (region, net) pairs of ONE grid shape (D, H, W) = (dim_z, dim_y, dim_x), straight from the access-point lists of the regions loaded in `b` — the 7 planes
 * of a net (baseline/build_3Dgrid.py:106-142) are never formed: they are 0/1 planes whose non-zeros are the net's access points, ResidualBlock(7) of such an input
 * differs from its response to an empty grid only within two voxels of one, and the aligning convolution is linear (xr_agent.hip, the NET branch).
 *   pair_region_dev / pair_net_dev  int32 [n_pairs]: region index in `b`, 1-based net id
 *   weights_dev   xr_agent_net_tower_weights() floats, bg_dev fp32 [od*oh*ow*7] = align1(block(0)) of this shape: both packed by agents.FusedNetTower
 *   out_dev       fp32 [n_pairs][64] (min-max normalised when normalize != 0, like the obstacle tower's rows)
 *   flags_dev     int32 [n_pairs], ZEROED BY THE CALLER: 1 = the net's neighbourhood lists do not fit LDS (more than ~20 scattered access points: the caller's
 *                 framework path takes that net), 2 = the pair names a region of another shape or a net it does not have; such rows of out_dev stay unwritten
 * XR_ERR_RANGE for grid shapes the kernel does not take. */
int32_t xr_agent_net_tower_weights(void);
/* How the towers' 7 -> 7-channel convolutions run (both kernels): 1 (default) = split-bf16 operands on v_mfma_f32_16x16x32_bf16 — every fp32 weight and activation held
 * as hi = bf16(v), lo = bf16(v - hi), a product as hi x hi + hi x lo + lo x hi accumulated in fp32 (normalised vectors within 1.1e-5 of the fp32 form on the reference
 * fixtures); 0 = fp32 operands on v_mfma_f32_16x16x4_f32 (environment XR_TOWER_FP32=1 before the first call).  Both weight layouts travel in the packed vectors. */
int32_t xr_agent_matrix_mode(void);
int32_t xr_batch_net_vectors(xr_batch* b, const int32_t* pair_region_dev, const int32_t* pair_net_dev, int32_t n_pairs, int32_t D, int32_t H, int32_t W,
                             const float* weights_dev, const float* bg_dev, float* out_dev, int32_t* flags_dev, int32_t normalize, void* stream);

/* ---- wire format (net_ordering.proto v1), host only ---------------------------------------- */
/* Replaces handle_messange's protobuf decode (baseline/baseline_utils.py:9-43; the `message.ParseFromString` of :418,467), with the
 * runtime's parse rules: the oneof keeps the LAST member on the wire, a repeated occurrence of the same member merges (scalars: last
 * value, `nodes` / `nets`: appended), unknown fields of every wire type (groups included) are skipped, packed and unpacked `nets` mix.
 * Pass 1 (fields_host == NULL, nets_host == NULL): fills info_host[8] = {kind (1 request, 2 response, 0 empty),
 * dim_x, dim_y, dim_z, n_nodes, n_nets, is_done, response.net_index} and metrics_host[3].
 * Pass 2: fields_host int32[fields_cap][10] = maze xyz, point xyz, type, is_used, net, pin (wire values, 0-based);
 * nets_host uint32[nets_cap].  Capacities are in rows / entries; nothing is ever written past them (ABI 9: round 5's two-pointer form
 * trusted pass 1's counts, which a oneof flip makes smaller than what an earlier, dropped member held).  XR_ERR_RANGE when what the
 * message finally holds does not fit (info_host carries the counts needed); XR_ERR_PARSE for bytes the runtimes refuse. */
int32_t xr_proto_decode(const uint8_t* buf_host, size_t len, int64_t* info_host, uint32_t* metrics_host,
                        int32_t* fields_host, int64_t fields_cap, uint32_t* nets_host, int64_t nets_cap);
/* Encodes Message{response{net_index}} exactly as Game.step does (:409-411). Returns bytes written
 * through *len (buf capacity >= 16). */
int32_t xr_proto_encode_response(int32_t net_index, uint8_t* buf_host, size_t* len);
/* Encodes Message{request{...}} (the simulator side; used by tests and the serve-the-protocol path).
 * Call with buf_host == NULL to get the required size in *len. */
int32_t xr_proto_encode_request(int32_t dim_x, int32_t dim_y, int32_t dim_z, const int32_t* fields_host,
                                int32_t n_nodes, const uint32_t* metrics_host, int32_t is_done,
                                const uint32_t* nets_host, int32_t n_nets, uint8_t* buf_host, size_t* len);

#ifdef __cplusplus
}
#endif
#endif /* XROUTE_HIP_H */
