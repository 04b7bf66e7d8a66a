/*
 * xr_oracle.c — CPU oracle for the xroute_env hot path.  TEST INFRASTRUCTURE ONLY (see xr_oracle.h:
 * who may load it, and the parity status: observation half pinned to tests/golden, router half
 * "parity unpinned" — it restates this repo's XR-Maze v1 spec).
 *
 * Plain C, no dependencies.  Written as a literal, unoptimised restatement: explicit obstacle and
 * access-point lists like the reference's Python, a textbook binary-heap Dijkstra for the router.
 * It deliberately shares no code and no algorithmic shortcut with the HIP path it checks.
 */
#include "xr_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define REC_TYPE(r) ((r) & 3u)
#define REC_USED(r) (((r) >> 2) & 1u)
#define REC_NET1(r) ((int)(((r) >> 3) & 0x3FFFu))
#define REC_PIN1(r) ((int)(((r) >> 17) & 0x3FFFu))
#define T_BLOCKAGE 0u
#define T_NORMAL 1u
#define T_ACCESS 2u
#define OWNER_FOREIGN 0x7FFF
#define INF32 0xFFFFFFFFu
#define DIST_CAP 0x07F00000u   /* XR-Maze v1: distances >= DIST_CAP do not exist (such a node is unreachable) */

#define ENV_BAD_ACTION 1
#define ENV_UNREACHABLE 2
#define ENV_PATH_TRUNC 4

/* The `data` triple handle_messange derives per node (reference baseline/baseline_utils.py:23-39):
 *   Net  = net+1 for ACCESS, -1 for BLOCKAGE, 0 for NORMAL;  bool_occupy = is_used.  */
static int rec_Net(uint32_t r) {
    if (REC_TYPE(r) == T_ACCESS) return REC_NET1(r);
    if (REC_TYPE(r) == T_BLOCKAGE) return -1;
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* netSet: keys of accessPoints after the training / inference filters.                           */
/* reference baseline/build_3Dgrid.py:18-55 (collect), :46-55 (training: drop routed nets),       */
/* :243-250 (inference: keep only nets listed in data[3]).                                        */
/* ------------------------------------------------------------------------------------------- */
static int contains(const int32_t* a, int n, int v) {
    for (int i = 0; i < n; i++) if (a[i] == v) return 1;
    return 0;
}

int xro_legal_nets(const uint32_t* rec, int n_nodes, const int32_t* routed, int n_routed,
                   const int32_t* filter, int n_filter, int inference, int32_t* out, int cap) {
    /* presence table over 1-based net ids */
    unsigned char* has = (unsigned char*)calloc(0x4000, 1);
    if (!has) return -1;
    for (int f = 0; f < n_nodes; f++) {
        int Net = rec_Net(rec[f]);
        if (Net >= 1) has[Net] = 1;                       /* :31-43 */
    }
    if (!inference) {
        for (int i = 0; i < n_routed; i++)                /* :46-55 */
            if (routed[i] >= 1 && routed[i] < 0x4000) has[routed[i]] = 0;
    } else {
        for (int net = 1; net < 0x4000; net++)            /* :243-250 */
            if (has[net] && !contains(filter, n_filter, net)) has[net] = 0;
    }
    int k = 0;
    for (int net = 1; net < 0x4000; net++)
        if (has[net]) { if (k < cap) out[k] = net; k++; }
    free(has);
    return k;                                             /* ascending == sorted(list(netSet)) :177 */
}

/* ------------------------------------------------------------------------------------------- */
/* observation.  Flat index of vertex (x,y,z) inside one channel is (x*Y+y)*Z+z because the       */
/* reference fills zeros([X,Y,Z])[x][y][z] and then reshapes (no permute) to [.., Z, Y, X]         */
/* (baseline/build_3Dgrid.py:97-103, :115-142).                                                    */
/* ------------------------------------------------------------------------------------------- */
int xro_build_observation(int X, int Y, int Z, const uint32_t* rec, const int32_t* nets, int k,
                          float* out) {
    const long N = (long)X * Y * Z;
    memset(out, 0, sizeof(float) * (size_t)(2 + 7 * (long)k) * (size_t)N);
    /* channel 0: obstacles (getObstaclesAndAccessPoints :19-36 + getObstacleGrid :94-103) */
    for (long f = 0; f < N; f++) {
        int Net = rec_Net(rec[f]);
        int occupy = (int)REC_USED(rec[f]);
        int obstacle = 0;
        if (Net == -1) obstacle = 1;                       /* :20-21 */
        else if (Net == 0) { if (occupy == 1) obstacle = 1; }   /* :22-25 */
        else { if (occupy == 1) obstacle = 1; }            /* :34-36 used AP is also an obstacle */
        if (obstacle) out[f] = 1.0f;
    }
    /* channel 1: net order channel, net ids at raster positions 0..k-1 (getNetOrderChannel :144-161) */
    for (int j = 0; j < k; j++) out[N + j] = (float)nets[j];
    /* 7 channels per net, nets ascending (_build_3Dgrid :177-179) */
    static const int dx[6] = {+1, 0, -1, 0, 0, 0};        /* east, south, west, north, up, down :127 */
    static const int dy[6] = {0, -1, 0, +1, 0, 0};        /* _get_adjacent_point :59-92 */
    static const int dz[6] = {0, 0, 0, 0, +1, -1};
    for (int i = 0; i < k; i++) {
        float* ch = out + (size_t)(2 + 7 * (long)i) * (size_t)N;
        const int net = nets[i];
        /* ch0: every access point of the net, any pin (getNetGrid :111-120) */
        for (long f = 0; f < N; f++)
            if (rec_Net(rec[f]) == net) ch[f] = 1.0f;
        /* ch1..6: `[t.zeros(grid_dim)] * 6` is ONE tensor listed six times (:125), so all six
         * channels receive every write of :138: AP has an in-bounds axis neighbour that is an AP of
         * the same net */
        for (int x = 0; x < X; x++) for (int y = 0; y < Y; y++) for (int z = 0; z < Z; z++) {
            long f = ((long)x * Y + y) * Z + z;
            if (rec_Net(rec[f]) != net) continue;
            for (int d = 0; d < 6; d++) {
                int ax = x + dx[d], ay = y + dy[d], az = z + dz[d];
                if (ax < 0 || ax >= X || ay < 0 || ay >= Y || az < 0 || az >= Z) continue;
                long g = ((long)ax * Y + ay) * Z + az;
                if (rec_Net(rec[g]) == net)
                    for (int c = 1; c <= 6; c++) ch[(size_t)c * (size_t)N + (size_t)f] = 1.0f;
            }
        }
    }
    return 0;
}

/* reference baseline/DQN/train_DQN.py:98-99: reward = -1; reward *= violation*500 + via*4 + wirelength*0.5
 * (Python: int*int exact, + float -> IEEE double) */
double xro_reward(int64_t v, int64_t w, int64_t via) {
    double s = (double)(v * 500 + via * 4) + (double)w * 0.5;
    return -1.0 * s;
}

/* ------------------------------------------------------------------------------------------- */
/* env                                                                                           */
/* ------------------------------------------------------------------------------------------- */
struct xro_env {
    int X, Y, Z, N, n_nets;
    int32_t *xs, *ys;
    uint8_t* ldir;
    uint32_t* rec0;        /* initial records */
    int32_t m0[3];
    int via_cost, pen_cost;
    /* XR-Maze v2 knobs, all neutral by default: guide_cost 0, maze_end_iter 1 */
    int guide_cost, guide_margin, maze_end_iter;
    int pen_now;           /* penalty of the current attempt: pen_cost << attempt */
    int n_gb; int gb[XRO_GUIDE_MAX_BOXES][6]; /* guide of the net being routed: boxes x0,x1,y0,y1,z0,z1 (track / layer indices, margin applied) */
    int32_t* guide_off; int16_t* guide_box;   /* optional per-net guide boxes (xro_env_set_guides): CSR over nets, 6 int16 per box */
    /* static derived */
    int16_t* node_net;     /* Net of handle_messange: -1 / 0 / net id (1-based) */
    int* net_off;          /* [n_nets+2] CSR into ap_node/ap_pin, by 1-based net id */
    int* ap_node;
    int* ap_pin;           /* pin+1 */
    /* mutable */
    int16_t* owner;        /* 0 free, net id, OWNER_FOREIGN */
    unsigned char* legal;  /* [n_nets+1] */
    int nlegal;
    int32_t cum[3];
    uint64_t hash;
    int64_t steps;
    /* scratch */
    uint32_t* dist;
    unsigned char* comp;
    int* heap_node; uint32_t* heap_key; int heap_n, heap_cap;
};

static void fnv_mix(uint64_t* h, uint32_t w) { *h = (*h ^ (uint64_t)w) * 0x100000001b3ULL; }

xro_env* xro_env_create(int X, int Y, int Z, const int32_t* xs, const int32_t* ys,
                        const uint8_t* layer_dir, const uint32_t* rec, int n_nets,
                        const int32_t* metrics0, int via_cost, int drc_cost, int drc_unit) {
    xro_env* e = (xro_env*)calloc(1, sizeof(xro_env));
    if (!e) return NULL;
    e->X = X; e->Y = Y; e->Z = Z; e->N = X * Y * Z; e->n_nets = n_nets;
    int N = e->N;
    e->xs = (int32_t*)malloc(sizeof(int32_t) * (X > 0 ? X : 1));
    e->ys = (int32_t*)malloc(sizeof(int32_t) * (Y > 0 ? Y : 1));
    e->ldir = (uint8_t*)malloc(Z > 0 ? Z : 1);
    e->rec0 = (uint32_t*)malloc(sizeof(uint32_t) * (N > 0 ? N : 1));
    memcpy(e->xs, xs, sizeof(int32_t) * X);
    memcpy(e->ys, ys, sizeof(int32_t) * Y);
    memcpy(e->ldir, layer_dir, Z);
    memcpy(e->rec0, rec, sizeof(uint32_t) * N);
    for (int i = 0; i < 3; i++) e->m0[i] = metrics0 ? metrics0[i] : 0;
    e->via_cost = via_cost;
    e->pen_cost = drc_cost * drc_unit;
    e->guide_cost = 0; e->guide_margin = 0; e->maze_end_iter = 1;
    e->pen_now = e->pen_cost;
    e->node_net = (int16_t*)malloc(sizeof(int16_t) * (N > 0 ? N : 1));
    e->owner = (int16_t*)malloc(sizeof(int16_t) * (N > 0 ? N : 1));
    e->legal = (unsigned char*)calloc(n_nets + 2, 1);
    e->dist = (uint32_t*)malloc(sizeof(uint32_t) * (N > 0 ? N : 1));
    e->comp = (unsigned char*)malloc(N > 0 ? N : 1);
    e->heap_cap = 8 * (N > 0 ? N : 1) + 16;
    e->heap_node = (int*)malloc(sizeof(int) * e->heap_cap);
    e->heap_key = (uint32_t*)malloc(sizeof(uint32_t) * e->heap_cap);
    /* CSR of access points per net, ordered by flat index (pins are looked up per AP) */
    e->net_off = (int*)calloc(n_nets + 2, sizeof(int));
    int total = 0;
    for (int f = 0; f < N; f++) {
        int Net = rec_Net(rec[f]);
        e->node_net[f] = (int16_t)Net;
        if (Net >= 1 && Net <= n_nets) { e->net_off[Net + 1]++; total++; }
    }
    for (int n = 1; n <= n_nets + 1; n++) e->net_off[n] += e->net_off[n - 1];
    e->ap_node = (int*)malloc(sizeof(int) * (total > 0 ? total : 1));
    e->ap_pin = (int*)malloc(sizeof(int) * (total > 0 ? total : 1));
    int* cur = (int*)malloc(sizeof(int) * (n_nets + 2));
    memcpy(cur, e->net_off, sizeof(int) * (n_nets + 2));
    for (int f = 0; f < N; f++) {
        int Net = e->node_net[f];
        if (Net >= 1 && Net <= n_nets) {
            e->ap_node[cur[Net]] = f;
            e->ap_pin[cur[Net]] = REC_PIN1(rec[f]);
            cur[Net]++;
        }
    }
    free(cur);
    e->hash = 0xcbf29ce484222325ULL;
    e->steps = 0;
    xro_env_reset(e);
    return e;
}

void xro_env_destroy(xro_env* e) {
    if (!e) return;
    free(e->xs); free(e->ys); free(e->ldir); free(e->rec0); free(e->node_net); free(e->owner);
    free(e->legal); free(e->dist); free(e->comp); free(e->heap_node); free(e->heap_key);
    free(e->net_off); free(e->ap_node); free(e->ap_pin); free(e->guide_off); free(e->guide_box);
    free(e);
}

/* Game.reset bookkeeping (reference baseline/baseline_utils.py:466-473): routed_nets = set(),
 * action_space = netSet of the initial Request, *_last_step = its cumulative metrics. */
void xro_env_reset(xro_env* e) {
    for (int f = 0; f < e->N; f++) {
        uint32_t r = e->rec0[f];
        int16_t ow = 0;
        if (REC_USED(r)) ow = (REC_TYPE(r) == T_ACCESS) ? (int16_t)REC_NET1(r) : (int16_t)OWNER_FOREIGN;
        e->owner[f] = ow;
    }
    e->nlegal = 0;
    for (int n = 1; n <= e->n_nets; n++) {
        e->legal[n] = (e->net_off[n + 1] > e->net_off[n]);   /* net has at least one AP (:31-43) */
        e->nlegal += e->legal[n];
    }
    for (int i = 0; i < 3; i++) e->cum[i] = e->m0[i];
}

int xro_env_nlegal(const xro_env* e) { return e->nlegal; }
int xro_env_n_nodes(const xro_env* e) { return e->N; }
int64_t xro_env_steps(const xro_env* e) { return e->steps; }
uint64_t xro_env_hash(const xro_env* e) { return e->hash; }
void xro_env_cum(const xro_env* e, int32_t cum[3]) { for (int i = 0; i < 3; i++) cum[i] = e->cum[i]; }
void xro_env_owner(const xro_env* e, int16_t* out) { memcpy(out, e->owner, sizeof(int16_t) * e->N); }

int xro_env_legal(const xro_env* e, int32_t* out, int cap) {
    int k = 0;
    for (int n = 1; n <= e->n_nets; n++)
        if (e->legal[n]) { if (k < cap) out[k] = n; k++; }
    return k;
}

/* current records: is_used reflects the owner array ("used by a net or is a blockage",
 * net_ordering.proto:24) */
static uint32_t cur_rec(const xro_env* e, int f) {
    uint32_t r = e->rec0[f] & ~4u;
    if (e->owner[f] != 0) r |= 4u;
    return r;
}

int xro_env_observation(const xro_env* e, float* out) {
    uint32_t* rec = (uint32_t*)malloc(sizeof(uint32_t) * (e->N > 0 ? e->N : 1));
    int32_t* nets = (int32_t*)malloc(sizeof(int32_t) * (e->n_nets + 1));
    for (int f = 0; f < e->N; f++) rec[f] = cur_rec(e, f);
    int k = xro_env_legal(e, nets, e->n_nets + 1);
    int rc = xro_build_observation(e->X, e->Y, e->Z, rec, nets, k, out);
    free(rec); free(nets);
    return rc;
}

/* ---- XR-Maze v1 --------------------------------------------------------- */
/* neighbour of v in direction d (0 E x+1, 1 S y-1, 2 W x-1, 3 N y+1, 4 U z+1, 5 D z-1) if the grid
 * graph has that edge: planar edges only along the layer's preferred direction, vias everywhere.
 * Returns -1 when there is no such edge; *len = edge length (DBU) or via cost. */
static int graph_nbr(const xro_env* e, int v, int d, uint32_t* len, int* is_via) {
    int Y = e->Y, Z = e->Z;
    int z = v % Z, y = (v / Z) % Y, x = v / (Y * Z);
    *is_via = 0;
    switch (d) {
    case 0: if (e->ldir[z] != 0 || x + 1 >= e->X) return -1; *len = (uint32_t)(e->xs[x + 1] - e->xs[x]); return v + Y * Z;
    case 2: if (e->ldir[z] != 0 || x - 1 < 0) return -1;     *len = (uint32_t)(e->xs[x] - e->xs[x - 1]); return v - Y * Z;
    case 1: if (e->ldir[z] != 1 || y - 1 < 0) return -1;     *len = (uint32_t)(e->ys[y] - e->ys[y - 1]); return v - Z;
    case 3: if (e->ldir[z] != 1 || y + 1 >= Y) return -1;    *len = (uint32_t)(e->ys[y + 1] - e->ys[y]); return v + Z;
    case 4: if (z + 1 >= Z) return -1; *len = (uint32_t)e->via_cost; *is_via = 1; return v + 1;
    default: if (z - 1 < 0) return -1; *len = (uint32_t)e->via_cost; *is_via = 1; return v - 1;
    }
}

static int node_blocked(const xro_env* e, int v) { return e->node_net[v] == -1; }
/* entering v while routing `net` costs the drc penalty and counts a violation when v is held by
 * another net's wire or is another net's access point */
static int node_pen(const xro_env* e, int v, int net) {
    int ow = e->owner[v], nn = e->node_net[v];
    return (ow != 0 && ow != net) || (nn > 0 && nn != net);
}

/* XR-Maze v2: entering a node outside the net's guide box costs guide_cost extra (no violation).  The guide of a net is the
 * bounding box of ALL its access points in track indices, inflated by guide_margin tracks in x and y, on every layer
 * (the role of `-follow_guide 1`, ispd/ispd18_test1/run-net-ordering-training.tcl:3: out-of-guide routing is expensive,
 * not forbidden). */
static uint32_t node_guide(const xro_env* e, int v) {
    if (e->guide_cost == 0) return 0u;
    int z = v % e->Z, y = (v / e->Z) % e->Y, x = v / (e->Y * e->Z);
    for (int i = 0; i < e->n_gb; i++) {
        const int* g = e->gb[i];
        if (x >= g[0] && x <= g[1] && y >= g[2] && y <= g[3] && z >= g[4] && z <= g[5]) return 0u;
    }
    return (uint32_t)e->guide_cost;
}

/* Optional: the global-route guide of every net as boxes (x0, y0, x1, y1, z0, z1; track / layer indices, inclusive), at most
 * XRO_GUIDE_MAX_BOXES per net: what `-follow_guide 1` reads from ispd18_test1.input.guide, clipped to the region
 * (xroute_env_amd/lefdef.py).  A node is inside the guide when it is inside any box inflated by guide_margin in x and y.
 * A net without boxes keeps the default guide (bounding box of its access points, every layer).  Returns 0, or -1 when a
 * net has too many boxes. */
int xro_env_set_guides(xro_env* e, const int32_t* box_off, const int16_t* boxes) {
    free(e->guide_off); free(e->guide_box); e->guide_off = NULL; e->guide_box = NULL;
    if (!box_off) return 0;
    for (int n = 0; n < e->n_nets; n++) if (box_off[n + 1] - box_off[n] > XRO_GUIDE_MAX_BOXES || box_off[n + 1] < box_off[n]) return -1;
    const int nb = box_off[e->n_nets];
    e->guide_off = (int32_t*)malloc(sizeof(int32_t) * (e->n_nets + 1));
    e->guide_box = (int16_t*)malloc(sizeof(int16_t) * 6 * (nb > 0 ? nb : 1));
    memcpy(e->guide_off, box_off, sizeof(int32_t) * (e->n_nets + 1));
    if (nb > 0) memcpy(e->guide_box, boxes, sizeof(int16_t) * 6 * nb);
    return 0;
}

void xro_env_set_v2(xro_env* e, int guide_cost, int guide_margin, int maze_end_iter) {
    e->guide_cost = guide_cost; e->guide_margin = guide_margin; e->maze_end_iter = maze_end_iter < 1 ? 1 : maze_end_iter;
}

static void heap_push(xro_env* e, uint32_t key, int node) {
    int i = e->heap_n++;
    while (i > 0) {
        int p = (i - 1) / 2;
        if (e->heap_key[p] <= key) break;
        e->heap_key[i] = e->heap_key[p]; e->heap_node[i] = e->heap_node[p];
        i = p;
    }
    e->heap_key[i] = key; e->heap_node[i] = node;
}
static int heap_pop(xro_env* e, uint32_t* key) {
    int top = e->heap_node[0];
    *key = e->heap_key[0];
    e->heap_n--;
    if (e->heap_n > 0) {
        uint32_t k = e->heap_key[e->heap_n]; int nd = e->heap_node[e->heap_n];
        int i = 0;
        for (;;) {
            int c = 2 * i + 1;
            if (c >= e->heap_n) break;
            if (c + 1 < e->heap_n && e->heap_key[c + 1] < e->heap_key[c]) c++;
            if (e->heap_key[c] >= k) break;
            e->heap_key[i] = e->heap_key[c]; e->heap_node[i] = e->heap_node[c];
            i = c;
        }
        e->heap_key[i] = k; e->heap_node[i] = nd;
    }
    return top;
}

/* multi-source Dijkstra from the component set; the distance field is unique */
static void dijkstra(xro_env* e, int net) {
    e->heap_n = 0;
    for (int v = 0; v < e->N; v++) {
        if (e->comp[v]) { e->dist[v] = 0; heap_push(e, 0, v); }
        else e->dist[v] = INF32;
    }
    while (e->heap_n > 0) {
        uint32_t du;
        int u = heap_pop(e, &du);
        if (du != e->dist[u]) continue;
        for (int d = 0; d < 6; d++) {
            uint32_t len; int is_via;
            int v = graph_nbr(e, u, d, &len, &is_via);
            if (v < 0 || node_blocked(e, v) || e->comp[v]) continue;
            uint64_t nd = (uint64_t)du + len + (node_pen(e, v, net) ? (uint32_t)e->pen_now : 0u) + node_guide(e, v);
            if (nd >= DIST_CAP) continue;   /* spec: such a node is unreachable */
            if ((uint32_t)nd < e->dist[v]) {
                e->dist[v] = (uint32_t)nd;
                if (e->heap_n < e->heap_cap) heap_push(e, (uint32_t)nd, v);
            }
        }
    }
}

int xro_env_distance_field(xro_env* e, int action, uint32_t* dist_out) {
    if (action < 1 || action > e->n_nets) return -1;
    int lo = e->net_off[action], hi = e->net_off[action + 1];
    if (hi <= lo) return -1;
    int first = 0x7FFFFFFF;
    for (int i = lo; i < hi; i++) if (e->ap_pin[i] < first) first = e->ap_pin[i];
    memset(e->comp, 0, e->N);
    for (int i = lo; i < hi; i++) if (e->ap_pin[i] == first) e->comp[e->ap_node[i]] = 1;
    dijkstra(e, action);
    memcpy(dist_out, e->dist, sizeof(uint32_t) * e->N);
    return 0;
}

int xro_env_step(xro_env* e, int action, int32_t delta[3], int* done, int32_t* path, int path_cap,
                 int* path_len) {
    int status = 0;
    int plen = 0;
    int32_t d_vio = 0, d_wl = 0, d_via = 0;
    delta[0] = delta[1] = delta[2] = 0;
    if (path_len) *path_len = 0;
    if (action < 1 || action > e->n_nets || !e->legal[action]) {
        if (done) *done = (e->nlegal == 0);
        return ENV_BAD_ACTION;
    }
    const int net = action;
    const int lo = e->net_off[net], hi = e->net_off[net + 1];
    /* distinct pins of the net */
    int npins = 0;
    int* pins = (int*)malloc(sizeof(int) * (hi - lo + 1));
    unsigned char* connected = (unsigned char*)calloc(hi - lo + 1, 1);
    for (int i = lo; i < hi; i++) {
        int p = e->ap_pin[i], seen = 0;
        for (int j = 0; j < npins; j++) if (pins[j] == p) { seen = 1; break; }
        if (!seen) pins[npins++] = p;
    }
    /* guide of the net (v2): its boxes when the region carries them, else the bounding box of its access points on every layer */
    e->n_gb = 0;
    if (e->guide_off && e->guide_off[net] > e->guide_off[net - 1]) {
        for (int i = e->guide_off[net - 1]; i < e->guide_off[net]; i++) {
            const int16_t* g = e->guide_box + 6 * (size_t)i;
            int* o = e->gb[e->n_gb++];
            o[0] = g[0] - e->guide_margin; o[1] = g[2] + e->guide_margin; o[2] = g[1] - e->guide_margin; o[3] = g[3] + e->guide_margin;
            o[4] = g[4]; o[5] = g[5];
        }
    } else {
        int gx0 = 1 << 30, gy0 = 1 << 30, gx1 = -1, gy1 = -1;
        for (int i = lo; i < hi; i++) {
            int f = e->ap_node[i], y = (f / e->Z) % e->Y, x = f / (e->Y * e->Z);
            if (x < gx0) gx0 = x;
            if (x > gx1) gx1 = x;
            if (y < gy0) gy0 = y;
            if (y > gy1) gy1 = y;
        }
        int* o = e->gb[e->n_gb++];
        o[0] = gx0 - e->guide_margin; o[1] = gx1 + e->guide_margin; o[2] = gy0 - e->guide_margin; o[3] = gy1 + e->guide_margin;
        o[4] = 0; o[5] = e->Z - 1;
    }
    /* XR-Maze v2 rip-up and reroute (`-maze_end_iter`, `-ripup_mode 1`): attempt t routes the whole net with the penalty
     * pen_cost << t; an attempt whose path uses a node held by another net is ripped up (owners restored, nothing
     * recorded) unless it is the last one.  maze_end_iter 1 = XR-Maze v1. */
    int16_t* owner0 = NULL;
    const uint64_t hash0 = e->hash;
    if (e->maze_end_iter > 1) { owner0 = (int16_t*)malloc(sizeof(int16_t) * (e->N > 0 ? e->N : 1)); memcpy(owner0, e->owner, sizeof(int16_t) * e->N); }
    for (int attempt = 0;; attempt++) {
    e->pen_now = e->pen_cost << attempt;
    int held_vio = 0;
    memset(connected, 0, hi - lo + 1);
    /* component starts as every access point of the lowest pin id */
    int first = pins[0], first_idx = 0;
    for (int j = 1; j < npins; j++) if (pins[j] < first) { first = pins[j]; first_idx = j; }
    memset(e->comp, 0, e->N);
    for (int i = lo; i < hi; i++) if (e->ap_pin[i] == first) e->comp[e->ap_node[i]] = 1;
    connected[first_idx] = 1;
    int remaining = npins - 1;

    while (remaining > 0) {
        dijkstra(e, net);
        /* nearest access point of a pin that is not connected yet; ties -> lowest flat index */
        int best = -1, best_pin_idx = -1;
        uint32_t best_d = INF32;
        for (int i = lo; i < hi; i++) {
            int pj = -1;
            for (int j = 0; j < npins; j++) if (pins[j] == e->ap_pin[i]) { pj = j; break; }
            if (connected[pj]) continue;
            int v = e->ap_node[i];
            uint32_t dv = e->dist[v];
            if (dv == INF32) continue;
            if (dv < best_d || (dv == best_d && v < best)) { best_d = dv; best = v; best_pin_idx = pj; }
        }
        if (best < 0) {           /* every remaining pin is unreachable */
            d_vio += remaining;
            status |= ENV_UNREACHABLE;
            break;
        }
        /* deterministic back-trace: first predecessor in the order E,S,W,N,U,D */
        int v = best;
        while (e->dist[v] > 0) {
            int pred = -1; uint32_t plen_e = 0; int pvia = 0;
            const int held = node_pen(e, v, net);     /* a violation even when drc_cost is 0 */
            uint32_t pen_v = (held ? (uint32_t)e->pen_now : 0u) + node_guide(e, v);
            for (int d = 0; d < 6; d++) {
                uint32_t len; int is_via;
                int u = graph_nbr(e, v, d, &len, &is_via);
                if (u < 0 || node_blocked(e, u) || e->dist[u] == INF32) continue;
                if ((uint64_t)e->dist[u] + len + pen_v == (uint64_t)e->dist[v]) {
                    pred = u; plen_e = len; pvia = is_via; break;
                }
            }
            if (pred < 0) { status |= 0x100; break; }   /* cannot happen on a consistent field */
            /* claim v */
            if (held) { d_vio += 1; held_vio += 1; }
            if (e->owner[v] == 0) e->owner[v] = (int16_t)net;
            e->comp[v] = 1;
            if (plen < path_cap && path) path[plen] = v;
            plen++;
            fnv_mix(&e->hash, (uint32_t)v);
            if (pvia) d_via += 1; else d_wl += (int32_t)plen_e;
            v = pred;
        }
        /* terminal node of the component: claimed (and recorded) only if nobody holds it yet */
        if (e->owner[v] == 0) {
            e->owner[v] = (int16_t)net;
            if (plen < path_cap && path) path[plen] = v;
            plen++;
            fnv_mix(&e->hash, (uint32_t)v);
        }
        /* the reached pin joins the component with all of its access points */
        connected[best_pin_idx] = 1;
        remaining--;
        for (int i = lo; i < hi; i++) if (e->ap_pin[i] == pins[best_pin_idx]) e->comp[e->ap_node[i]] = 1;
    }
    if (held_vio == 0 || attempt + 1 >= e->maze_end_iter) break;
    /* rip up */
    memcpy(e->owner, owner0, sizeof(int16_t) * e->N);
    e->hash = hash0;
    status = 0; plen = 0; d_vio = 0; d_wl = 0; d_via = 0;
    }
    e->pen_now = e->pen_cost;
    free(owner0);
    free(pins); free(connected);
    if (plen > path_cap) status |= ENV_PATH_TRUNC;

    /* Game.step bookkeeping (reference baseline/baseline_utils.py:412, :426-438): the simulator
     * reports cumulative metrics, the env returns deltas; the routed net leaves netSet; done when
     * netSet is empty */
    e->cum[0] += d_vio; e->cum[1] += d_wl; e->cum[2] += d_via;
    delta[0] = d_vio; delta[1] = d_wl; delta[2] = d_via;
    e->legal[net] = 0;
    e->nlegal--;
    if (done) *done = (e->nlegal == 0);
    if (path_len) *path_len = plen;
    fnv_mix(&e->hash, (uint32_t)action);
    fnv_mix(&e->hash, (uint32_t)d_vio); fnv_mix(&e->hash, (uint32_t)d_wl); fnv_mix(&e->hash, (uint32_t)d_via);
    fnv_mix(&e->hash, (uint32_t)plen);
    e->steps++;
    return status;
}

/* ------------------------------------------------------------------------------------------- */
/* batch helpers                                                                                 */
/* ------------------------------------------------------------------------------------------- */
int xro_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

void xro_batch_random_actions(xro_env** envs, int n, uint64_t seed, int32_t* actions) {
    for (int i = 0; i < n; i++) {
        xro_env* e = envs[i];
        if (e->nlegal == 0) { actions[i] = 0; continue; }
        uint64_t r = splitmix64(seed ^ splitmix64((uint64_t)i * 0x100000001B3ULL + (uint64_t)e->steps));
        int j = (int)(r % (uint64_t)e->nlegal);
        for (int net = 1; net <= e->n_nets; net++)
            if (e->legal[net]) { if (j == 0) { actions[i] = net; break; } j--; }
    }
}

int64_t xro_batch_step(xro_env** envs, const int32_t* actions, int n, int threads, int auto_reset,
                       int32_t* delta, uint8_t* done, double* reward) {
    int64_t real = 0;
    if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4) reduction(+ : real)
#endif
    for (int i = 0; i < n; i++) {
        xro_env* e = envs[i];
        int32_t d[3] = {0, 0, 0};
        int dn = 0;
        if (auto_reset && e->nlegal == 0) {
            xro_env_reset(e);
            dn = (e->nlegal == 0);
        } else {
            int st = xro_env_step(e, actions[i], d, &dn, NULL, 0, NULL);
            if (!(st & ENV_BAD_ACTION)) real++;
        }
        if (delta) { delta[3 * i] = d[0]; delta[3 * i + 1] = d[1]; delta[3 * i + 2] = d[2]; }
        if (done) done[i] = (uint8_t)dn;
        if (reward) reward[i] = xro_reward(d[0], d[1], d[2]);
    }
    return real;
}

int xro_batch_observation(xro_env** envs, int n, float* out, int64_t stride, int threads) {
    if (threads < 1) threads = 1;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4)
#endif
    for (int i = 0; i < n; i++) xro_env_observation(envs[i], out + (size_t)i * (size_t)stride);
    return 0;
}
