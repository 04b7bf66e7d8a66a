/* dial_sim.c — CPU model of the bucketed-frontier router (development tool, not the oracle, not shipped):
 * same XR-Maze v1 spec, the relaxation organised as the GPU kernel does it (rounds over an open bitmask,
 * bucket width mult * w_min, bound pruning + deferred re-open, field re-used between the searches of one net).
 * Prints statistics and returns path/metrics so that a harness can compare with oracle/xr_oracle.c. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define CAP 0x07F00000u
#define UNREACHED 0xFFFFFFFDu

int g_astar = 0;      /* 1: buckets keyed on f = d + h(v), h = distance to the bounding box of the unconnected targets */
int g_chain = 0;      /* 1: a neighbour lowered into the current bucket is expanded within the same round */
typedef struct {
    long rounds, expansions, relax_ok, searches, max_open, scans_nonempty_words, routes;
} dial_stats;

/* returns 0; out: delta[3], path (cap), plen */
int dial_route(int X, int Y, int Z, const int32_t* xs, const int32_t* ys, const uint8_t* ldir,
               const int16_t* node_net, int16_t* owner, int net, const int32_t* ap_node, const int32_t* ap_pin, int nap,
               int via_cost, int pen_cost, int mult, int32_t* delta, int32_t* path, int path_cap, int* plen_out,
               dial_stats* st) {
    const int N = X * Y * Z, YZ = Y * Z;
    uint32_t* field = (uint32_t*)malloc(4 * (size_t)N);
    uint8_t* open = (uint8_t*)calloc(N, 1);
    uint8_t* defer = (uint8_t*)calloc(N, 1);
    uint8_t* conn = (uint8_t*)calloc(nap, 1);
    for (int f = 0; f < N; f++) {
        int nn = node_net[f], ow = owner[f];
        if (nn == -1) field[f] = 0;
        else field[f] = UNREACHED | (((ow != 0 && ow != net) || (nn > 0 && nn != net)) ? 2u : 0u);
    }
    uint32_t wmin = (uint32_t)via_cost;
    for (int i = 1; i < X; i++) if ((uint32_t)(xs[i] - xs[i - 1]) < wmin) wmin = xs[i] - xs[i - 1];
    for (int i = 1; i < Y; i++) if ((uint32_t)(ys[i] - ys[i - 1]) < wmin) wmin = ys[i] - ys[i - 1];
    const uint32_t delta_w = wmin * (uint32_t)mult;
    int first = 0x7fffffff, npins = 0;
    for (int i = 0; i < nap; i++) {
        if (ap_pin[i] < first) first = ap_pin[i];
        int seen = 0;
        for (int j = 0; j < i; j++) if (ap_pin[j] == ap_pin[i]) seen = 1;
        npins += !seen;
    }
    for (int i = 0; i < nap; i++) if (ap_pin[i] == first) { conn[i] = 1; field[ap_node[i]] &= 3u; open[ap_node[i]] = 1; }
    int remaining = npins - 1, plen = 0, d_vio = 0, d_wl = 0, d_via = 0;
    st->routes++;
    while (remaining > 0) {
        st->searches++;
        /* bounding box of the unconnected targets (coordinates) */
        int bx0 = 1 << 30, bx1 = -(1 << 30), by0 = 1 << 30, by1 = -(1 << 30), bz0 = 1 << 30, bz1 = -1;
        for (int i = 0; i < nap; i++) if (!conn[i]) {
            int f = ap_node[i], x = f / YZ, y = (f / Z) % Y, z = f % Z;
            if (xs[x] < bx0) bx0 = xs[x]; if (xs[x] > bx1) bx1 = xs[x];
            if (ys[y] < by0) by0 = ys[y]; if (ys[y] > by1) by1 = ys[y];
            if (z < bz0) bz0 = z; if (z > bz1) bz1 = z;
        }
        /* per-pin boxes (g_astar == 2): h = min over the unconnected pins of the distance to the pin's own box */
        int npb = 0; int pbx0[64], pbx1[64], pby0[64], pby1[64], pbz0[64], pbz1[64], pbpin[64];
        for (int i = 0; i < nap; i++) if (!conn[i]) {
            int k = -1; for (int q = 0; q < npb; q++) if (pbpin[q] == ap_pin[i]) k = q;
            if (k < 0 && npb < 64) { k = npb++; pbpin[k] = ap_pin[i]; pbx0[k] = pby0[k] = pbz0[k] = 1 << 30; pbx1[k] = pby1[k] = -(1 << 30); pbz1[k] = -1; }
            if (k < 0) continue;
            int f = ap_node[i], x = f / YZ, y = (f / Z) % Y, z = f % Z;
            if (xs[x] < pbx0[k]) pbx0[k] = xs[x]; if (xs[x] > pbx1[k]) pbx1[k] = xs[x];
            if (ys[y] < pby0[k]) pby0[k] = ys[y]; if (ys[y] > pby1[k]) pby1[k] = ys[y];
            if (z < pbz0[k]) pbz0[k] = z; if (z > pbz1[k]) pbz1[k] = z;
        }
#define HEUR2(f) ({ int _x = (f) / YZ, _y = ((f) / Z) % Y, _z = (f) % Z; uint32_t _best = 0xFFFFFFFFu; \
            for (int _q = 0; _q < npb; _q++) { \
            int _hx = xs[_x] < pbx0[_q] ? pbx0[_q] - xs[_x] : (xs[_x] > pbx1[_q] ? xs[_x] - pbx1[_q] : 0); \
            int _hy = ys[_y] < pby0[_q] ? pby0[_q] - ys[_y] : (ys[_y] > pby1[_q] ? ys[_y] - pby1[_q] : 0); \
            int _hz = _z < pbz0[_q] ? pbz0[_q] - _z : (_z > pbz1[_q] ? _z - pbz1[_q] : 0); \
            uint32_t _h = _hx + _hy + _hz * via_cost; if (_h < _best) _best = _h; } _best == 0xFFFFFFFFu ? 0u : _best; })
#define HEUR(f) (g_astar == 2 ? HEUR2(f) : g_astar ? (uint32_t)(({ int _x = (f) / YZ, _y = ((f) / Z) % Y, _z = (f) % Z; \
            int _hx = xs[_x] < bx0 ? bx0 - xs[_x] : (xs[_x] > bx1 ? xs[_x] - bx1 : 0); \
            int _hy = ys[_y] < by0 ? by0 - ys[_y] : (ys[_y] > by1 ? ys[_y] - by1 : 0); \
            int _hz = _z < bz0 ? bz0 - _z : (_z > bz1 ? _z - bz1 : 0); _hx + _hy + _hz * via_cost; })) : 0u)
        for (int f = 0; f < N; f++) if (defer[f]) { open[f] = 1; defer[f] = 0; }
        uint32_t best = 0xFFFFFFFFu;
        for (;;) {
            for (int i = 0; i < nap; i++) if (!conn[i]) { uint32_t w = field[ap_node[i]]; if (w < UNREACHED && (w >> 2) < best) best = w >> 2; }
            uint32_t m = 0xFFFFFFFFu; long nopen = 0;
            for (int f = 0; f < N; f++) if (open[f]) { nopen++; uint32_t k = (field[f] >> 2) + HEUR(f); if (k < m) m = k; }
            if (nopen > st->max_open) st->max_open = nopen;
            if (m == 0xFFFFFFFFu || m > best) break;
            st->rounds++;
            const uint32_t hi = m + delta_w;
            /* snapshot semantics of one parallel round: collect the bucket first */
            int* bucket = (int*)malloc(sizeof(int) * (nopen + 1)); int nb = 0;
            int bcap = nopen + 1;
            for (int f = 0; f < N; f++) if (open[f] && (field[f] >> 2) + HEUR(f) < hi) { bucket[nb++] = f; open[f] = 0; }
            for (int bi = 0; bi < nb; bi++) {
                const int f = bucket[bi];
                const uint32_t w = field[f], d = w >> 2;
                st->expansions++;
                const int x = f / YZ, y = (f / Z) % Y, z = f % Z;
                int nf[4]; uint32_t len[4]; int cnt = 0;
                if (!ldir[z]) {
                    if (x + 1 < X) { nf[cnt] = f + YZ; len[cnt++] = xs[x + 1] - xs[x]; }
                    if (x > 0) { nf[cnt] = f - YZ; len[cnt++] = xs[x] - xs[x - 1]; }
                } else {
                    if (y + 1 < Y) { nf[cnt] = f + Z; len[cnt++] = ys[y + 1] - ys[y]; }
                    if (y > 0) { nf[cnt] = f - Z; len[cnt++] = ys[y] - ys[y - 1]; }
                }
                if (z + 1 < Z) { nf[cnt] = f + 1; len[cnt++] = via_cost; }
                if (z > 0) { nf[cnt] = f - 1; len[cnt++] = via_cost; }
                for (int k = 0; k < cnt; k++) {
                    const uint32_t wn = field[nf[k]];
                    if (wn == 0) continue;
                    const uint64_t cand = (uint64_t)d + len[k] + ((wn & 2u) ? (uint32_t)pen_cost : 0u);
                    if (cand >= CAP) continue;
                    const uint32_t cw = ((uint32_t)cand << 2) | (wn & 3u);
                    if (cw >= wn) continue;
                    if (cand + HEUR(nf[k]) > best) { defer[f] = 1; continue; }
                    field[nf[k]] = cw; st->relax_ok++;
                    if (g_chain && cand + HEUR(nf[k]) < hi) {
                        if (nb >= bcap) { bcap *= 2; bucket = (int*)realloc(bucket, sizeof(int) * bcap); }
                        bucket[nb++] = nf[k]; open[nf[k]] = 0;
                    } else open[nf[k]] = 1;
                }
            }
            free(bucket);
        }
        /* target */
        int bi = -1; uint32_t bd = 0xFFFFFFFFu; int bf = 0;
        for (int i = 0; i < nap; i++) {
            if (conn[i]) continue;
            uint32_t w = field[ap_node[i]];
            if (w >= UNREACHED) continue;
            if ((w >> 2) < bd || ((w >> 2) == bd && ap_node[i] < bf)) { bd = w >> 2; bf = ap_node[i]; bi = i; }
        }
        if (bi < 0) { d_vio += remaining; break; }
        int v = bf;
        int* claimed = (int*)malloc(sizeof(int) * N); int nc = 0;
        while ((field[v] >> 2) > 0) {
            const uint32_t vw = field[v];
            const uint32_t need = (vw >> 2) - ((vw & 2u) ? (uint32_t)pen_cost : 0u);
            const int x = v / YZ, y = (v / Z) % Y, z = v % Z;
            int u = -1, via = 0; uint32_t ul = 0;
            for (int d = 0; d < 6 && u < 0; d++) {
                int c = -1; uint32_t len = 0;
                switch (d) {
                case 0: if (!ldir[z] && x + 1 < X) { c = v + YZ; len = xs[x + 1] - xs[x]; } break;
                case 1: if (ldir[z] && y > 0) { c = v - Z; len = ys[y] - ys[y - 1]; } break;
                case 2: if (!ldir[z] && x > 0) { c = v - YZ; len = xs[x] - xs[x - 1]; } break;
                case 3: if (ldir[z] && y + 1 < Y) { c = v + Z; len = ys[y + 1] - ys[y]; } break;
                case 4: if (z + 1 < Z) { c = v + 1; len = via_cost; } break;
                default: if (z > 0) { c = v - 1; len = via_cost; } break;
                }
                if (c < 0) continue;
                const uint32_t cw = field[c];
                if (cw == 0 || cw >= UNREACHED) continue;
                if ((cw >> 2) + len == need) { u = c; via = d >= 4; ul = len; }
            }
            if (u < 0) { free(claimed); free(field); free(open); free(defer); free(conn); return -1; }
            if (vw & 2u) d_vio++;
            claimed[nc++] = v;
            if (plen < path_cap) path[plen] = v;
            plen++;
            if (via) d_via++; else d_wl += (int)ul;
            v = u;
        }
        if (owner[v] == 0) { owner[v] = (int16_t)net; if (plen < path_cap) path[plen] = v; plen++; }
        for (int i = 0; i < nc; i++) { field[claimed[i]] &= 3u; open[claimed[i]] = 1; if (owner[claimed[i]] == 0) owner[claimed[i]] = (int16_t)net; }
        free(claimed);
        const int pin = ap_pin[bi];
        for (int i = 0; i < nap; i++) if (ap_pin[i] == pin) { conn[i] = 1; field[ap_node[i]] &= 3u; open[ap_node[i]] = 1; }
        remaining--;
    }
    delta[0] = d_vio; delta[1] = d_wl; delta[2] = d_via; *plen_out = plen;
    free(field); free(open); free(defer); free(conn);
    return 0;
}
