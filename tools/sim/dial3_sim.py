"""CPU model of the round-3 frontier router ("v3", xroute_env_amd/csrc/xr_dial3.h) — the design was validated here, against
the oracle, before the kernel existed.  Test infrastructure / design notebook only: nothing in the product imports it.

What is modelled (everything that could change a RESULT; timing-only choices such as lane counts are not):
  * field word  = dist << 5 | pdir << 2 | held << 1 | valid.  Lowering = min() on the WORD, so among candidates of equal
    distance the lowest predecessor direction (E,S,W,N,U,D = 0..5: the spec's back-trace order) wins — the back-trace becomes
    a pointer chase and still picks "the first tight predecessor in the order E,S,W,N,U,D";
  * the field persists over the searches of one route (new sources: path nodes + the reached pin's access points);
  * A* keys d + h, h = distance to the bounding box of the access points of the still unconnected pins;
  * bucket grid of width 2^dshift on the keys; the current bucket's queue, a `later` list of (node, bucket tag) whose tags
    are LOWER BOUNDS (h only grows from one search to the next), lazy re-bucketing when an entry is popped;
  * bound pruning: a candidate with key > best is not written, its parent goes to the `defer` list (tag = the bucket it was
    expanded in) and is looked at again by the next search;
  * list capacities with the spill set as the fallback (run with tiny capacities to exercise it).

    python tools/sim/dial3_sim.py [n_regions] [seed0] [cap]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

INF_W = 0xFFFFFFFD
DIRS = ((1, 0, 0), (0, -1, 0), (-1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, -1))       # E S W N U D
OPP = (2, 3, 0, 1, 5, 4)


class V3:
    def __init__(self, region, via=800, pen=3200, cap_cur=128, cap_later=512, hot_window=6, dshift=None, nq=16):
        self.r = region
        self.X, self.Y, self.Z = (int(v) for v in region.dims)
        self.N = self.X * self.Y * self.Z
        rec = region.nodes.astype(np.int64)
        t = rec & 3
        net1 = (rec >> 3) & 0x3FFF
        self.pin1 = (rec >> 17) & 0x3FFF
        self.node_net = np.where(t == 2, net1, np.where(t == 0, -1, 0))
        used = (rec >> 2) & 1
        self.owner = np.where(used == 1, np.where(t == 2, net1, 0x7FFF), 0).astype(np.int64)
        self.xs = [int(v) for v in region.xs]
        self.ys = [int(v) for v in region.ys]
        self.ld = [int(v) for v in region.layer_dir]
        self.via, self.pen = via, pen
        wmin = via
        for i in range(1, self.X):
            wmin = min(wmin, self.xs[i] - self.xs[i - 1])
        for i in range(1, self.Y):
            wmin = min(wmin, self.ys[i] - self.ys[i - 1])
        self.dshift = dshift if dshift else max(11, int(round(np.log2(8 * max(1, wmin)))))
        self.caps = (cap_cur, cap_later, hot_window)
        self.nq = nq
        self.stats = dict(expansions=0, rounds=0, pops=0, stale=0, rebucket=0, spills=0)

    def xyz(self, f):
        YZ = self.Y * self.Z
        return f // YZ, (f // self.Z) % self.Y, f % self.Z

    def nbr(self, f, d):
        """neighbour of f in direction d (0..5) along the graph, its edge length — or None"""
        x, y, z = self.xyz(f)
        dx, dy, dz = DIRS[d]
        if dz == 0:
            if dx and self.ld[z] != 0:
                return None
            if dy and self.ld[z] != 1:
                return None
        nx, ny, nz = x + dx, y + dy, z + dz
        if not (0 <= nx < self.X and 0 <= ny < self.Y and 0 <= nz < self.Z):
            return None
        ln = abs(self.xs[nx] - self.xs[x]) + abs(self.ys[ny] - self.ys[y]) if dz == 0 else self.via
        return (nx * self.Y + ny) * self.Z + nz, ln

    def route(self, a):
        N = self.N
        aps = [f for f in range(N) if self.node_net[f] == a]
        pins = sorted(set(int(self.pin1[f]) for f in aps))
        first = pins[0]
        W = np.zeros(N, np.int64)
        for f in range(N):
            if self.node_net[f] == -1:
                W[f] = 0
            else:
                held = (self.owner[f] != 0 and self.owner[f] != a) or (self.node_net[f] > 0 and self.node_net[f] != a)
                W[f] = INF_W | (2 if held else 0)
        conn = {p: (p == first) for p in pins}
        capC, capL, hotw = self.caps
        # Sets (node bitmasks in the kernel): `open` = queued (hot list / current queue) and not expanded since — the duplicate
        # filter of a pop; `cold` = lowered, not expanded, in no list (leftovers of earlier searches, overflow, keys beyond the hot
        # window), cold_lb = a lower bound of its keys (keys only grow from one search to the next: h does); `defer` = an edge out
        # of the node was refused by the bound.
        open_set, cold_set, defer_set = set(), set(), set()
        cold = dict(lb=None)
        newsrc = [f for f in aps if int(self.pin1[f]) == first]
        for f in newsrc:
            W[f] &= 3
        remaining = len(pins) - 1
        path_all, d_vio, d_wl, d_via, status = [], 0, 0, 0, 0
        st = self.stats
        while remaining > 0:
            tg = [f for f in aps if not conn[int(self.pin1[f])]]
            cx = [self.xs[self.xyz(f)[0]] for f in tg]
            cy = [self.ys[self.xyz(f)[1]] for f in tg]
            cz = [self.xyz(f)[2] for f in tg]
            hb = (min(cx), max(cx), min(cy), max(cy), min(cz), max(cz))

            def heur(f):
                x, y, z = self.xyz(f)
                xc, yc = self.xs[x], self.ys[y]
                return max(0, hb[0] - xc, xc - hb[1]) + max(0, hb[2] - yc, yc - hb[3]) + max(0, hb[4] - z, z - hb[5]) * self.via

            def bkt(key):
                return key >> self.dshift

            later = []
            state = dict(base=0)

            def to_cold(f, key):
                cold_set.add(f)
                cold["lb"] = key if cold["lb"] is None else min(cold["lb"], key)

            def push(f, key):
                b = bkt(key)
                if b < state["base"] + hotw and len(later) < capL:
                    open_set.add(f)
                    later.append((f, b))
                else:
                    to_cold(f, key)

            def classify(nodes):
                st["classify"] = st.get("classify", 0) + 1
                st["classified"] = st.get("classified", 0) + len(nodes)
                for f in nodes:
                    push(f, (int(W[f]) >> 5) + heur(f))
                st["hot_max"] = max(st.get("hot_max", 0), len(later))

            # search start: the new sources and the deferred nodes are classified; the cold set stays cold
            srcs = sorted(set(newsrc) | defer_set)
            st["defer_n"] = st.get("defer_n", 0) + len(defer_set)
            defer_set.clear()
            if srcs:
                state["base"] = bkt(min((int(W[f]) >> 5) + heur(f) for f in srcs))
            classify(srcs)
            best = None
            while True:
                # ---- bucket advance -----------------------------------------------------------------------------------------
                tvals = [int(W[f] >> 5) for f in tg if (int(W[f]) | 2) != 0xFFFFFFFF]
                best = min(tvals) if tvals else None
                bnew = min((t for _, t in later), default=None)
                # lower bound of every key that is still to be expanded: the search is over when it exceeds best
                lbs = ([bnew << self.dshift] if bnew is not None else []) + ([cold["lb"]] if cold_set else [])
                if not lbs or (best is not None and min(lbs) > best):
                    break
                if cold_set and (bnew is None or (bkt(cold["lb"]) <= bnew and len(later) <= capL // 2)):
                    # the frontier reached the cold set's lower bound (and the hot list has room): classify the cold set — what stays
                    # cold gets an exact lower bound.  No room: the hot bucket goes first (order never affects the result).
                    st["cold_scans"] = st.get("cold_scans", 0) + 1
                    nodes = sorted(cold_set)
                    cold_set.clear()
                    cold["lb"] = None
                    keys = [(int(W[f]) >> 5) + heur(f) for f in nodes]
                    state["base"] = min(bkt(min(keys)), bnew if bnew is not None else 1 << 30)
                    classify(nodes)
                    continue
                if bnew is None:
                    break
                bcur = bnew
                st["rounds"] += 1
                st["part_entries"] = st.get("part_entries", 0) + len(later)
                hi = (bcur + 1) << self.dshift
                cur = [f for f, t in later if t == bcur]
                later[:] = [(f, t) for f, t in later if t != bcur]
                if len(cur) > capC:
                    for f in cur[capC:]:
                        open_set.discard(f)
                        to_cold(f, bcur << self.dshift)
                    cur = cur[:capC]
                # ---- the wave's hop loop in lockstep: NQ quads; an idle quad pops the next queue entry, an active one does one hop ----
                NQ = self.nq
                quad = [None] * NQ
                qh = 0
                while True:
                    for q in range(NQ):                      # idle quads pop (in quad order, as the kernel's ballot/prefix does)
                        while quad[q] is None and qh < len(cur):
                            g = cur[qh]
                            qh += 1
                            st["pops"] += 1
                            if g not in open_set:             # the entry is a duplicate: the node was expanded since
                                st["stale"] += 1
                                break                         # (the kernel's quad stays idle for this iteration)
                            open_set.discard(g)
                            quad[q] = g
                    if all(v is None for v in quad):
                        if qh >= len(cur):
                            break
                        continue
                    st["hop_iters"] = st.get("hop_iters", 0) + 1
                    # all active quads read their neighbours first (lockstep), then the atomics land in lane order
                    plans = []
                    for q in range(NQ):
                        g = quad[q]
                        if g is None:
                            continue
                        st["expansions"] += 1
                        gd = int(W[g]) >> 5
                        x, y, z = self.xyz(g)
                        vert = self.ld[z] == 1
                        for d in ((3 if vert else 0), (1 if vert else 2), 4, 5):
                            nb = self.nbr(g, d)
                            if nb is None:
                                continue
                            nf, ln = nb
                            wn = int(W[nf])
                            if wn == 0:
                                continue
                            cand = gd + ln + (self.pen if (wn & 2) else 0)
                            cw = (cand << 5) | (OPP[d] << 2) | (wn & 3)
                            if not (cand < (1 << 27) - 64 and cw < wn):
                                continue
                            k = cand + heur(nf)
                            plans.append((q, nf, cw, k, best is not None and k > best))
                    wins = {}
                    for q, nf, cw, k, refused in plans:
                        if refused:
                            defer_set.add(quad[q])
                            continue
                        old = int(W[nf])
                        if cw < old:
                            W[nf] = cw
                            if (cw >> 5) < (old >> 5):            # the distance really went down
                                if q not in wins and k < hi:
                                    wins[q] = nf
                                elif bkt(k) == bcur and len(cur) < capC:
                                    open_set.add(nf)
                                    cur.append(nf)
                                else:
                                    push(nf, k)
                    for q in range(NQ):
                        if quad[q] is not None:
                            quad[q] = wins.get(q)
            # search end: what is still queued goes cold (its keys are >= bcur's lower edge, or beyond best)
            for f, t in later:
                if f in open_set:
                    open_set.discard(f)
                    to_cold(f, t << self.dshift)
            assert not open_set, open_set
            # ---- target: nearest access point of an unconnected pin, ties -> lowest flat index --------------------------
            cands = [(int(W[f]) >> 5, f) for f in tg if (int(W[f]) | 2) != 0xFFFFFFFF]
            if not cands:
                d_vio += remaining
                status |= 2
                break
            dist, tf = min(cands)
            # ---- back-trace = pointer chase along pdir ------------------------------------------------------------------
            v = tf
            newsrc = []
            while (int(W[v]) >> 5) > 0:
                w = int(W[v])
                pd = (w >> 2) & 7
                u, _ = self.nbr(v, pd)
                wu = int(W[u])
                held = bool(w & 2)
                ln = (w >> 5) - (wu >> 5) - (self.pen if held else 0)
                if held:
                    d_vio += 1
                if pd >= 4:
                    d_via += 1
                else:
                    d_wl += ln
                path_all.append(v)
                newsrc.append(v)
                v = u
                assert len(newsrc) <= N
            for f in newsrc:
                W[f] &= 3
                if self.owner[f] == 0:
                    self.owner[f] = a
            if self.owner[v] == 0:
                self.owner[v] = a
                path_all.append(v)
            pin = int(self.pin1[tf])
            conn[pin] = True
            remaining -= 1
            for f in aps:
                if int(self.pin1[f]) == pin:
                    W[f] &= 3
                    newsrc.append(f)
        return dict(path=path_all, delta=[d_vio, d_wl, d_via], status=status)


    def route_k(self, a):
        N = self.N
        aps = [f for f in range(N) if self.node_net[f] == a]
        pins = sorted(set(int(self.pin1[f]) for f in aps))
        first = pins[0]
        W = np.zeros(N, np.int64)
        for f in range(N):
            if self.node_net[f] == -1:
                W[f] = 0
            else:
                held = (self.owner[f] != 0 and self.owner[f] != a) or (self.node_net[f] > 0 and self.node_net[f] != a)
                W[f] = INF_W | (2 if held else 0)
        conn = {p: (p == first) for p in pins}
        capC, capL, hotw = self.caps
        # Sets (node bitmasks in the kernel): `open` = queued (hot list / current queue) and not expanded since — the duplicate
        # filter of a pop; `cold` = lowered, not expanded, in no list (leftovers of earlier searches, overflow, keys beyond the hot
        # window), cold_lb = a lower bound of its keys (keys only grow from one search to the next: h does); `defer` = an edge out
        # of the node was refused by the bound.
        open_set, cold_set, defer_set = set(), set(), set()
        cold = dict(lb=None)
        newsrc = [f for f in aps if int(self.pin1[f]) == first]
        for f in newsrc:
            W[f] &= 3
        remaining = len(pins) - 1
        path_all, d_vio, d_wl, d_via, status = [], 0, 0, 0, 0
        st = self.stats
        while remaining > 0:
            tg = [f for f in aps if not conn[int(self.pin1[f])]]
            cx = [self.xs[self.xyz(f)[0]] for f in tg]
            cy = [self.ys[self.xyz(f)[1]] for f in tg]
            cz = [self.xyz(f)[2] for f in tg]
            hb = (min(cx), max(cx), min(cy), max(cy), min(cz), max(cz))

            def heur(f):
                x, y, z = self.xyz(f)
                xc, yc = self.xs[x], self.ys[y]
                return max(0, hb[0] - xc, xc - hb[1]) + max(0, hb[2] - yc, yc - hb[3]) + max(0, hb[4] - z, z - hb[5]) * self.via

            def bkt(key):
                return key >> self.dshift

            later = []
            state = dict(base=0, hotlim=0x10000)

            def to_cold(f, key):
                cold_set.add(f)
                cold["lb"] = key if cold["lb"] is None else min(cold["lb"], key)

            def push(f, key):
                b = bkt(key)
                if b < state["hotlim"] and len(later) < capL:
                    later.append((f, b))
                else:
                    to_cold(f, key)

            def classify(nodes):
                st["classify"] = st.get("classify", 0) + 1
                st["classified"] = st.get("classified", 0) + len(nodes)
                for f in nodes:
                    push(f, (int(W[f]) >> 5) + heur(f))
                st["hot_max"] = max(st.get("hot_max", 0), len(later))

            # search start: the new sources and the deferred nodes are classified; the cold set stays cold
            srcs = sorted(set(newsrc) | defer_set)
            st["defer_n"] = st.get("defer_n", 0) + len(defer_set)
            defer_set.clear()
            if srcs:
                state["base"] = bkt(min((int(W[f]) >> 5) + heur(f) for f in srcs))
            classify(srcs)
            best = None
            while True:
                # ---- bucket advance -----------------------------------------------------------------------------------------
                tvals = [int(W[f] >> 5) for f in tg if (int(W[f]) | 2) != 0xFFFFFFFF]
                best = min(tvals) if tvals else None
                bnew = min((t for _, t in later), default=None)
                # lower bound of every key that is still to be expanded: the search is over when it exceeds best
                lbs = ([bnew << self.dshift] if bnew is not None else []) + ([cold["lb"]] if cold_set else [])
                if not lbs or (best is not None and min(lbs) > best):
                    break
                if cold_set and (bnew is None or (bkt(cold["lb"]) <= bnew and len(later) <= capL // 2)):
                    # the frontier reached the cold set's lower bound (and the hot list has room): classify the cold set — what stays
                    # cold gets an exact lower bound.  No room: the hot bucket goes first (order never affects the result).
                    st["cold_scans"] = st.get("cold_scans", 0) + 1
                    cold_lb0 = cold["lb"]
                    nodes = sorted(cold_set)
                    cold_set.clear()
                    cold["lb"] = None
                    keys = [(int(W[f]) >> 5) + heur(f) for f in nodes]
                    cb = bkt(cold_lb0)
                    state["hotlim"] = (bnew if (bnew is not None and bnew < cb) else cb) + hotw
                    classify(nodes)
                    continue
                if bnew is None:
                    break
                bcur = bnew
                state["hotlim"] = bnew + hotw
                st["rounds"] += 1
                st["part_entries"] = st.get("part_entries", 0) + len(later)
                hi = (bcur + 1) << self.dshift
                cur = [f for f, t in later if t == bcur]
                later[:] = [(f, t) for f, t in later if t != bcur]
                if len(cur) > capC:
                    later.extend((f, bcur) for f in cur[capC:])
                    cur = cur[:capC]
                # ---- the wave's hop loop in lockstep: NQ quads; an idle quad pops the next queue entry, an active one does one hop ----
                NQ = self.nq
                quad = [None] * NQ
                qh = 0
                while True:
                    for q in range(NQ):                      # idle quads pop (in quad order, as the kernel's ballot/prefix does)
                        while quad[q] is None and qh < len(cur):
                            g = cur[qh]
                            qh += 1
                            st["pops"] += 1
                            quad[q] = g
                    if all(v is None for v in quad):
                        if qh >= len(cur):
                            break
                        continue
                    st["hop_iters"] = st.get("hop_iters", 0) + 1
                    # all active quads read their neighbours first (lockstep), then the atomics land in lane order
                    plans = []
                    for q in range(NQ):
                        g = quad[q]
                        if g is None:
                            continue
                        st["expansions"] += 1
                        gd = int(W[g]) >> 5
                        x, y, z = self.xyz(g)
                        vert = self.ld[z] == 1
                        for d in ((3 if vert else 0), (1 if vert else 2), 4, 5):
                            nb = self.nbr(g, d)
                            if nb is None:
                                continue
                            nf, ln = nb
                            wn = int(W[nf])
                            if wn == 0:
                                continue
                            cand = gd + ln + (self.pen if (wn & 2) else 0)
                            cw = (cand << 5) | (OPP[d] << 2) | (wn & 3)
                            if not (cand < (1 << 27) - 64 and cw < wn):
                                continue
                            k = cand + heur(nf)
                            plans.append((q, nf, cw, k, best is not None and k > best))
                    wins = {}
                    for q, nf, cw, k, refused in plans:
                        if refused:
                            defer_set.add(quad[q])
                            continue
                        old = int(W[nf])
                        if cw < old:
                            W[nf] = cw
                            if (cw >> 5) < (old >> 5):            # the distance really went down
                                if q not in wins and k < hi:
                                    wins[q] = nf
                                elif bkt(k) == bcur and len(cur) < capC:
                                    cur.append(nf)
                                else:
                                    push(nf, k)
                    for q in range(NQ):
                        if quad[q] is not None:
                            quad[q] = wins.get(q)
            # search end: what is still queued goes cold (its keys are >= bcur's lower edge, or beyond best)
            for f, t in later:
                to_cold(f, t << self.dshift)
            # ---- target: nearest access point of an unconnected pin, ties -> lowest flat index --------------------------
            cands = [(int(W[f]) >> 5, f) for f in tg if (int(W[f]) | 2) != 0xFFFFFFFF]
            if not cands:
                d_vio += remaining
                status |= 2
                break
            dist, tf = min(cands)
            # ---- back-trace = pointer chase along pdir ------------------------------------------------------------------
            v = tf
            newsrc = []
            while (int(W[v]) >> 5) > 0:
                w = int(W[v])
                pd = (w >> 2) & 7
                u, _ = self.nbr(v, pd)
                wu = int(W[u])
                held = bool(w & 2)
                ln = (w >> 5) - (wu >> 5) - (self.pen if held else 0)
                if held:
                    d_vio += 1
                if pd >= 4:
                    d_via += 1
                else:
                    d_wl += ln
                path_all.append(v)
                newsrc.append(v)
                v = u
                assert len(newsrc) <= N
            for f in newsrc:
                W[f] &= 3
                if self.owner[f] == 0:
                    self.owner[f] = a
            if self.owner[v] == 0:
                self.owner[v] = a
                path_all.append(v)
            pin = int(self.pin1[tf])
            conn[pin] = True
            remaining -= 1
            for f in aps:
                if int(self.pin1[f]) == pin:
                    W[f] &= 3
                    newsrc.append(f)
        return dict(path=path_all, delta=[d_vio, d_wl, d_via], status=status)


def main():
    from oracle import xr_oracle as orc
    from xroute_env_amd.regions import generate_region
    nreg = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    cap = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    rng = np.random.default_rng(11)
    routes = bad = 0
    tot = dict(expansions=0, rounds=0, pops=0, stale=0, classify=0, classified=0, cold_scans=0, part_entries=0, defer_n=0)
    mx = dict(hot_max=0, open_max=0)
    for i in range(nreg):
        reg = generate_region(seed0 + i)
        env = orc.OracleEnv(reg)
        sim = V3(reg, **(dict(cap_cur=cap, cap_later=cap, hot_window=2) if cap else {}))
        while env.nlegal():
            a = int(rng.choice(env.legal()))
            ref = env.step(a)
            got = sim.route(a)
            routes += 1
            ok = got["path"] == ref["path"].tolist() and got["delta"] == ref["delta"].tolist() and (got["status"] & 2) == (ref["status"] & 2)
            if not ok:
                bad += 1
                print("MISMATCH region", i, "net", a, got["delta"], ref["delta"].tolist(), len(got["path"]), ref["path_len"])
            assert np.array_equal(sim.owner[:env.n].astype(np.int16), env.owner()) or not ok
        for k in tot:
            tot[k] += sim.stats.get(k, 0)
        for k in mx:
            mx[k] = max(mx[k], sim.stats.get(k, 0))
    print(f"{routes} routes, {bad} mismatches; per route:", {k: round(v / max(routes, 1), 2) for k, v in tot.items()}, mx)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
