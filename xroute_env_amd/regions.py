"""Region model + deterministic synthetic region generator (host side, numpy).

A *region* is what the reference receives from the simulator as one protobuf
``Request`` (reference: baseline/openroad_api/proto/net_ordering.proto:29-45): the maze
dimensions, one record per maze node (maze index, real coordinates, type, is_used,
net, pin), the cumulative metrics and the list of nets still to route.  Here it is
held as dense SoA arrays in the reference observation's own flat order

    f = (x * Y + y) * Z + z          (reference: baseline/build_3Dgrid.py:97-103, the
                                       zeros([X,Y,Z]) tensor that is reshaped, not permuted)

so that every device sweep over a region is a unit-stride sweep.

Packed node record (uint32), shared with include/xroute_hip.h (XR_REC_*):

    bits  1:0   node type   0 BLOCKAGE, 1 NORMAL, 2 ACCESS   (proto enum NodeType)
    bit   2     is_used
    bits 16:3   net + 1     (0 = no net; net ids are 0-based on the wire)
    bits 30:17  pin + 1     (0 = no pin)

ispd18_test1-sized regions (reference ispd/ispd18_test1/ispd18_test1.input.def:
234-251): X tracks pitch 400 DBU, Y tracks the union of the 380 / 570 / 760 DBU families,
9 routing layers, Metal1 horizontal and alternating (ispd18_test1.input.lef:13-196).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

BLOCKAGE, NORMAL, ACCESS = 0, 1, 2
NET_BITS = 14
MAX_NET = (1 << NET_BITS) - 2  # net+1 must fit in 14 bits


def pack_records(ntype, used, net, pin) -> np.ndarray:
    """Pack per-node fields into the uint32 record described in the module docstring."""
    ntype = np.asarray(ntype, dtype=np.int64)
    used = np.asarray(used, dtype=np.int64)
    net = np.asarray(net, dtype=np.int64)
    pin = np.asarray(pin, dtype=np.int64)
    if net.size and (net.max(initial=-1) > MAX_NET or pin.max(initial=-1) > MAX_NET):
        raise ValueError("net/pin id exceeds the 14-bit record field")
    rec = (ntype & 3) | ((used & 1) << 2) | (((net + 1) & 0x3FFF) << 3) | (((pin + 1) & 0x3FFF) << 17)
    return rec.astype(np.uint32)


def records_from_entries(n: int, flat, Net, used, Pin) -> np.ndarray:
    """Dense packed records from a node LIST that may name a vertex more than once, with the meaning the reference gives such a
    list (baseline/build_3Dgrid.py:18-43 looks at every entry on its own): the vertex is an obstacle if ANY of its entries is
    (Net == -1, or occupied), and an access point of net n if ANY of its entries says so.  `Net` follows handle_messange:
    -1 blockage, 0 plain node, >= 1 the 1-based net of an access point; `Pin` 1-based or -1.  In dense form an obstacle that is also
    an access point is a used ACCESS node — the same observation.  Not representable (ValueError): one vertex as an access point
    of two different nets.  One vertex listed under two different PINS of the same net (the reference appends it to both pins' lists,
    :37-43; its observation only asks "access point of this net, any pin", :126-138, so the tensor is the same) keeps the LOWEST pin
    in the one-pin-per-node record — deterministic whatever the list order; only the build-defined router sees pins.
    Vertices the list does not mention are unused NORMAL nodes."""
    flat = np.asarray(flat, np.int64)
    Net = np.asarray(Net, np.int64)
    used = np.asarray(used, np.int64) == 1
    Pin = np.asarray(Pin, np.int64)
    ntype = np.full(n, NORMAL, np.int64)
    u = np.zeros(n, np.int64)
    net = np.full(n, -1, np.int64)
    pin = np.full(n, -1, np.int64)
    if len(flat):
        acc, blk = Net >= 1, Net == -1
        any_acc = np.zeros(n, bool); any_blk = np.zeros(n, bool); any_used = np.zeros(n, bool)
        np.logical_or.at(any_acc, flat, acc)
        np.logical_or.at(any_blk, flat, blk)
        np.logical_or.at(any_used, flat, used)
        lo = np.full(n, np.iinfo(np.int64).max, np.int64); hi = np.full(n, -1, np.int64)
        np.minimum.at(lo, flat[acc], Net[acc])
        np.maximum.at(hi, flat[acc], Net[acc])
        if (any_acc & (lo != hi)).any():
            raise ValueError("a vertex is listed as an access point of two different nets: not representable as one node record")
        ntype[any_blk] = BLOCKAGE
        ntype[any_acc] = ACCESS
        u[any_used | (any_blk & any_acc)] = 1
        net[flat[acc]] = Net[acc] - 1
        # lowest pin over the entries that NAME one (Pin >= 1); -1 only when no entry of the vertex does
        big = np.iinfo(np.int64).max
        plo = np.full(n, big, np.int64)
        named = acc & (Pin >= 1)
        np.minimum.at(plo, flat[named], Pin[named] - 1)
        pin[any_acc] = np.where(plo[any_acc] == big, -1, plo[any_acc])
    return pack_records(ntype, u, net, pin)


def unpack_records(rec: np.ndarray):
    rec = np.asarray(rec, dtype=np.uint32).astype(np.int64)
    ntype = rec & 3
    used = (rec >> 2) & 1
    net = ((rec >> 3) & 0x3FFF) - 1
    pin = ((rec >> 17) & 0x3FFF) - 1
    return ntype, used, net, pin


@dataclass
class Region:
    """One routing region (= one simulator ``Request``), dense SoA form."""

    dims: tuple                      # (X, Y, Z)
    xs: np.ndarray                   # int32[X]  track x coordinates (DBU), strictly increasing
    ys: np.ndarray                   # int32[Y]
    layer_dir: np.ndarray            # uint8[Z]  0 = horizontal (x moves), 1 = vertical (y moves)
    nodes: np.ndarray                # uint32[N] packed records, flat order f=(x*Y+y)*Z+z
    n_nets: int                      # nets are 0..n_nets-1 on the wire (1-based in the env API)
    metrics0: np.ndarray = field(default_factory=lambda: np.zeros(3, np.int32))  # cum. (vio, wl, via)
    name: str = ""
    # optional global-route guides (XR-Maze v2 `guide_cost`): boxes (x0, y0, x1, y1, z0, z1) in track / layer indices, inclusive,
    # CSR over the nets (guide_off int32[n_nets + 1], guide_box int16[n_boxes, 6], at most 8 per net); None = the default guide
    guide_off: Optional[np.ndarray] = None
    guide_box: Optional[np.ndarray] = None

    @property
    def n_nodes(self) -> int:
        return int(self.dims[0]) * int(self.dims[1]) * int(self.dims[2])

    def flat(self, x, y, z):
        X, Y, Z = self.dims
        return (np.asarray(x) * Y + np.asarray(y)) * Z + np.asarray(z)

    def unflat(self, f):
        X, Y, Z = self.dims
        f = np.asarray(f)
        return f // (Y * Z), (f // Z) % Y, f % Z

    # ---- conversion to / from the reference's `data` nested list --------------------------
    def to_reference_data(self, nets: Optional[Sequence[int]] = None, metrics=None,
                          nodes: Optional[np.ndarray] = None) -> list:
        """The `data` list handle_messange would build (reference:
        baseline/baseline_utils.py:15-40): [[X,Y,Z], [[maze],[point],[used,type,pin]]*N,
        [vio,wl,via], nets(1-based)].  Used only by tests / fixture generation."""
        X, Y, Z = self.dims
        rec = self.nodes if nodes is None else nodes
        ntype, used, net, pin = unpack_records(rec)
        xs, ys, zs = self.unflat(np.arange(self.n_nodes))
        node_type = np.where(ntype == ACCESS, net + 1, np.where(ntype == BLOCKAGE, -1, 0))
        node_pin = np.where(ntype == ACCESS, pin + 1, -1)
        m = self.metrics0 if metrics is None else metrics
        if nets is None:
            nets = list(range(self.n_nets))
        items = []
        for i in range(self.n_nodes):
            x, y, z = int(xs[i]), int(ys[i]), int(zs[i])
            items.append([[x, y, z], [int(self.xs[x]), int(self.ys[y]), z],
                          [int(used[i]), int(node_type[i]), int(node_pin[i])]])
        return [[X, Y, Z], items, [int(m[0]), int(m[1]), int(m[2])], [int(n) + 1 for n in nets]]


def region_from_reference_data(data: list, layer_dir: Optional[Sequence[int]] = None) -> Region:
    """Dense Region from the reference's `data` list (any node order; nodes that are
    absent from the list become unused NORMAL nodes, which the reference also ignores:
    baseline/build_3Dgrid.py:18-43 only acts on listed vertices)."""
    X, Y, Z = (int(v) for v in data[0])
    n = X * Y * Z
    ntype = np.full(n, NORMAL, np.int64)
    used = np.zeros(n, np.int64)
    net = np.full(n, -1, np.int64)
    pin = np.full(n, -1, np.int64)
    xs = np.arange(X, dtype=np.int64)
    ys = np.arange(Y, dtype=np.int64)
    have_x = np.zeros(X, bool)
    have_y = np.zeros(Y, bool)
    if len(data[1]):
        maze = np.array([v[0] for v in data[1]], dtype=np.int64).reshape(-1, 3)
        point = np.array([v[1] for v in data[1]], dtype=np.int64).reshape(-1, 3)
        info = np.array([v[2] for v in data[1]], dtype=np.int64).reshape(-1, 3)
        if (maze < 0).any() or (maze[:, 0] >= X).any() or (maze[:, 1] >= Y).any() or (maze[:, 2] >= Z).any():
            raise ValueError("maze index outside the region dimensions")
        f = (maze[:, 0] * Y + maze[:, 1]) * Z + maze[:, 2]
        t = info[:, 1]
        ntype[f] = np.where(t == -1, BLOCKAGE, np.where(t == 0, NORMAL, ACCESS))
        used[f] = info[:, 0] != 0
        net[f] = np.where(t >= 1, t - 1, -1)
        pin[f] = np.where(t >= 1, info[:, 2] - 1, -1)
        xs[maze[:, 0]] = point[:, 0]
        ys[maze[:, 1]] = point[:, 1]
        have_x[maze[:, 0]] = True
        have_y[maze[:, 1]] = True
    xs = _monotone_fill(xs, have_x)
    ys = _monotone_fill(ys, have_y)
    if layer_dir is None:
        layer_dir = [z & 1 for z in range(Z)]
    k = int(net.max(initial=-1)) + 1
    return Region((X, Y, Z), xs.astype(np.int32), ys.astype(np.int32),
                  np.asarray(layer_dir, np.uint8), pack_records(ntype, used, net, pin), k,
                  np.asarray(data[2], np.int32))


def _monotone_fill(c: np.ndarray, have: np.ndarray) -> np.ndarray:
    """Coordinates for tracks no listed node mentions: keep the array strictly increasing."""
    c = c.astype(np.int64).copy()
    if not have.any():
        return np.arange(len(c), dtype=np.int64)
    idx = np.flatnonzero(have)
    if not (np.diff(c[idx]) > 0).all():
        return np.arange(len(c), dtype=np.int64)  # degenerate (test inputs): unit pitch
    out = np.interp(np.arange(len(c)), idx, c[idx]).round().astype(np.int64)
    first, last = idx[0], idx[-1]
    out[:first] = c[first] - (first - np.arange(first))
    out[last + 1:] = c[last] + (np.arange(last + 1, len(c)) - last)
    for i in range(1, len(out)):
        if out[i] <= out[i - 1]:
            out[i] = out[i - 1] + 1
    return out


# --------------------------------------------------------------------------------------------
# synthetic generator
# --------------------------------------------------------------------------------------------

def ispd18_tracks(rng: np.random.Generator, X: int, Y: int):
    """Track coordinates of an ispd18_test1 window: X tracks at pitch 400 (offset 200), Y tracks
    the union of 190+380i, 760+570j and 950+760k (ispd18_test1.input.def:234-251)."""
    x0 = 200 + 400 * int(rng.integers(0, 900))
    xs = x0 + 400 * np.arange(X, dtype=np.int64)
    ybase = int(rng.integers(0, 300000))
    span = 760 * (Y + 4)
    fam = np.concatenate([
        190 + 380 * np.arange((ybase + span) // 380 + 2),
        760 + 570 * np.arange((ybase + span) // 570 + 2),
        950 + 760 * np.arange((ybase + span) // 760 + 2)])
    fam = np.unique(fam[fam >= ybase])
    ys = fam[:Y]
    assert len(ys) == Y
    return xs.astype(np.int32), ys.astype(np.int32)


def generate_region(seed: int, dims=(24, 40, 9), k_range=(4, 36), blockage=(0.10, 0.20),
                    prerouted=(0.02, 0.06), pins=(2, 4), aps=(1, 4), net_span=10,
                    used_ap_prob=0.03, name: str = "") -> Region:
    """Deterministic synthetic region (SURVEY.md §8d "Configs as synthetic inputs").

    * blockage: runs of 1..6 nodes along the layer's preferred direction, denser on the
      low layers, until `blockage` fraction of the nodes is BLOCKAGE;
    * pre-routed occupancy: runs of used NORMAL nodes (wires of nets outside the region);
    * nets: K ~ U[k_range], 2..4 pins each inside a local window of `net_span` tracks,
      every pin 1..4 axis-adjacent ACCESS nodes on z in {0, 1}.
    """
    rng = np.random.default_rng(seed)
    X, Y, Z = dims
    n = X * Y * Z
    xs, ys = ispd18_tracks(rng, X, Y)
    layer_dir = (np.arange(Z) & 1).astype(np.uint8)

    ntype = np.full(n, NORMAL, np.int64)
    used = np.zeros(n, np.int64)
    net = np.full(n, -1, np.int64)
    pin = np.full(n, -1, np.int64)

    def flat(x, y, z):
        return (x * Y + y) * Z + z

    # per-layer weights: low layers are denser (cell pins / obstructions live there)
    lw = np.linspace(1.6, 0.5, Z)
    lw = lw / lw.sum()

    def paint_runs(frac: float):
        target = int(frac * n)
        out = np.zeros(n, bool)
        guard = 0
        while out.sum() < target and guard < 64:
            guard += 1
            m = max(8, (target - int(out.sum())) // 3 + 1)
            z = rng.choice(Z, size=m, p=lw)
            x = rng.integers(0, X, size=m)
            y = rng.integers(0, Y, size=m)
            ln = rng.integers(1, 7, size=m)
            for s in range(6):
                live = ln > s
                xx = np.where(layer_dir[z] == 0, x + s, x)
                yy = np.where(layer_dir[z] == 1, y + s, y)
                ok = live & (xx < X) & (yy < Y)
                out[flat(xx[ok], yy[ok], z[ok])] = True
                if out.sum() >= target:
                    break
        return out

    blk = paint_runs(float(rng.uniform(*blockage)))
    ntype[blk] = BLOCKAGE
    used[blk] = 1                      # proto: is_used is true for blockages too (net_ordering.proto:24)
    pre = paint_runs(float(rng.uniform(*prerouted))) & ~blk
    used[pre] = 1

    K = int(rng.integers(k_range[0], k_range[1] + 1))
    K = max(1, min(K, MAX_NET))
    taken = blk | pre
    for k in range(K):
        npin = int(rng.integers(pins[0], pins[1] + 1))
        cx = int(rng.integers(0, X))
        cy = int(rng.integers(0, Y))
        span = int(rng.integers(3, net_span + 1))
        placed = 0
        tries = 0
        while placed < npin and tries < 200:
            tries += 1
            px = int(np.clip(cx + rng.integers(-span, span + 1), 0, X - 1))
            py = int(np.clip(cy + rng.integers(-span, span + 1), 0, Y - 1))
            pz = int(rng.integers(0, min(2, Z)))
            nap = int(rng.integers(aps[0], aps[1] + 1))
            axis = int(rng.integers(0, 3))
            cells = []
            for a in range(nap):
                qx, qy, qz = px, py, pz
                if axis == 0:
                    qx += a
                elif axis == 1:
                    qy += a
                else:
                    qz += a
                if qx >= X or qy >= Y or qz >= min(2, Z):
                    break
                f = flat(qx, qy, qz)
                if taken[f]:
                    break
                cells.append(f)
            if not cells:
                continue
            for f in cells:
                taken[f] = True
                ntype[f] = ACCESS
                net[f] = k
                pin[f] = placed
                used[f] = 1 if rng.random() < used_ap_prob else 0
            placed += 1
        if placed == 0:
            # could not place anything: drop the remaining nets so ids stay dense
            K = k
            break
    m0 = np.array([int(rng.integers(0, 4)), int(rng.integers(0, 20000)), int(rng.integers(0, 20))],
                  np.int32)
    return Region((X, Y, Z), xs, ys, layer_dir, pack_records(ntype, used, net, pin), K, m0,
                  name or f"synth{seed}")


def generate_regions(n: int, base_seed: int = 1000, **kw) -> List[Region]:
    return [generate_region(base_seed + i, **kw) for i in range(n)]


# BASELINE.json configs -> generator arguments.  seed = 1000*config + env id.
CONFIGS = {
    1: dict(dims=(24, 40, 9), k_range=(10, 10)),
    2: dict(dims=(24, 40, 9), k_range=(4, 36)),
    3: dict(dims=(24, 40, 9), k_range=(4, 36)),
    4: dict(dims=(24, 40, 9), k_range=(4, 36)),
    5: dict(dims=(256, 256, 12), k_range=(32, 32), blockage=(0.25, 0.35), prerouted=(0.05, 0.10),
            net_span=48),
}


def config_regions(config: int, n: int, first_env: int = 0) -> List[Region]:
    kw = CONFIGS[config]
    return [generate_region(1000 * config + first_env + i, **kw) for i in range(n)]
