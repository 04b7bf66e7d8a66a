/*
 * xr_oracle.h — CPU oracle for the xroute_env hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker / the timed CPU baseline.  The product (xroute_env_amd/) never links,
 * imports or calls it.
 *
 * Parity status
 *   - observation / legal-set / metric-delta / reward half (reference baseline/build_3Dgrid.py,
 *     baseline/baseline_utils.py, baseline/DQN/train_DQN.py:98-99): PINNED against fixtures that
 *     tools/gen_golden.py produced by importing the reference (tests/golden/g1..g4).
 *   - router half (reference: external OpenROAD/TritonRoute binary, xplanlab/OpenROAD fork,
 *     version unpinned, source absent from the reference tree; call sites
 *     ispd/ispd18_test1/run-net-ordering-training.tcl:1-10, examples/launch_training.py:57-86):
 *     PARITY UNPINNED.  The router here restates this repository's own published spec
 *     "XR-Maze v1" (DESIGN.md §3), not TritonRoute.
 */
#ifndef XR_ORACLE_H
#define XR_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct xro_env xro_env;

/* ---- reference-pinned half ------------------------------------------------------------- */
/* netSet of build_3Dgrid (reference baseline/build_3Dgrid.py:6-56 + :243-250). Output ascending. */
int xro_legal_nets(const uint32_t* rec, int n_nodes, const int32_t* routed, int n_routed,
                   const int32_t* filter, int n_filter, int inference, int32_t* out, int cap);
/* observation [2+7k, Z, Y, X] fp32 (reference baseline/build_3Dgrid.py:94-188). */
int xro_build_observation(int X, int Y, int Z, const uint32_t* rec, const int32_t* nets, int k,
                          float* out);
/* reference baseline/DQN/train_DQN.py:98-99, baseline/PPO/train_PPO.py:101-102 */
double xro_reward(int64_t d_violation, int64_t d_wirelength, int64_t d_via);

/* ---- env (Game bookkeeping + XR-Maze v1 router) ------------------------------------------ */
xro_env* xro_env_create(int X, int Y, int Z, const int32_t* xs, const int32_t* ys,
                        const uint8_t* layer_dir, const uint32_t* rec, int n_nets,
                        const int32_t* metrics0, int via_cost, int drc_cost, int drc_unit);
/* XR-Maze v2 knobs: guide cost / margin, rip-up-and-reroute attempts; (0, 0, 1) = XR-Maze v1 */
void xro_env_set_v2(xro_env* e, int guide_cost, int guide_margin, int maze_end_iter);
/* XR-Maze v2, optional: the nets' global-route guides as boxes (x0, y0, x1, y1, z0, z1: track / layer indices, inclusive), CSR over
 * the nets (box_off[n_nets + 1], 0-based), at most XRO_GUIDE_MAX_BOXES per net; NULL = none (guide = bbox of the access points) */
#define XRO_GUIDE_MAX_BOXES 8
int xro_env_set_guides(xro_env* e, const int32_t* box_off, const int16_t* boxes);
void xro_env_destroy(xro_env* e);
void xro_env_reset(xro_env* e);   /* reference Game.reset bookkeeping, baseline_utils.py:466-473 */
/* reference Game.step bookkeeping (baseline_utils.py:409-438) around the XR-Maze v1 router.
 * action is 1-based. path: flat node ids claimed, back-trace order. Returns XR_ENV_* status bits. */
int xro_env_step(xro_env* e, int action, int32_t delta[3], int* done, int32_t* path, int path_cap,
                 int* path_len);
int xro_env_nlegal(const xro_env* e);
int xro_env_legal(const xro_env* e, int32_t* out, int cap);     /* ascending 1-based ids */
void xro_env_cum(const xro_env* e, int32_t cum[3]);
void xro_env_owner(const xro_env* e, int16_t* out);
uint64_t xro_env_hash(const xro_env* e);
int xro_env_observation(const xro_env* e, float* out);            /* current state, current netSet */
int xro_env_n_nodes(const xro_env* e);
/* distance field of a single-net search, for kernel unit tests: sources = APs of the lowest pin
 * of `action`; dist_out uint32[N] (0xFFFFFFFF = unreachable) */
int xro_env_distance_field(xro_env* e, int action, uint32_t* dist_out);

/* ---- batch helpers (OpenMP over envs) for the timed CPU baseline ------------------------ */
int xro_max_threads(void);
/* one env-step per env; done envs are reset instead (vector-env autoreset) when auto_reset.
 * returns the number of real env-steps performed */
int64_t xro_batch_step(xro_env** envs, const int32_t* actions, int n, int threads, int auto_reset,
                       int32_t* delta /*[n][3]*/, uint8_t* done /*[n]*/, double* reward /*[n]*/);
/* observation of every env into out + i*stride (floats) */
int xro_batch_observation(xro_env** envs, int n, float* out, int64_t stride, int threads);
/* pick action = legal[ hash(seed, i, step) % nlegal ] exactly as xr_batch_random_actions */
void xro_batch_random_actions(xro_env** envs, int n, uint64_t seed, int32_t* actions);
int64_t xro_env_steps(const xro_env* e);

#ifdef __cplusplus
}
#endif
#endif
