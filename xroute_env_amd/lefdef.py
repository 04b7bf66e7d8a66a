"""LEF / DEF / route-guide reader and per-GCell region extractor (SURVEY.md §8 f3 — a "next" row).

The reference gets its regions from a TritonRoute worker dump inside the external OpenROAD binary
(`detailed_route_run_worker -dump_dir ...`, ispd/ispd18_test1/run-net-ordering-training.tcl:9-10); the inputs of
that flow are the LEF / DEF / guide triple it ships (ispd/ispd18_test1/ispd18_test1.input.{lef,def,guide},
simulator/testcases/ispd18_sample/*).  This module turns the same three files into `Region` descriptors without the
binary.  There is NO oracle for this step (pin access is TritonRoute's own algorithm): the rules below are
build-defined and documented, and the output is only claimed to be a self-consistent, design-derived workload.

Rules
  * maze grid of a region = every (x track) x (y track) x (routing layer) crossing inside the region box, tracks
    taken from the DEF `TRACKS` statements of all routing layers (def:234-251 for ispd18_test1);
  * a node is BLOCKAGE when a wire centred on it would touch a macro OBS rectangle, a power/ground pin or a signal
    pin that is not routed in this region (those shapes are grown by width/2 + spacing of the layer); ACCESS (net, pin) when it lies in a pin rectangle of a net routed here; a pin without any
    on-grid node is snapped to the nearest node of its layer within `snap` DBU, else dropped;
  * a net whose global-route guide leaves the region gets one *boundary pin* per guide rectangle crossing the region
    edge on a layer whose preferred direction crosses that edge (one ACCESS node on the edge, at the middle of the
    crossing) — the role TritonRoute's boundary pins play;
  * a net is routed in the region when it has at least two pins (local or boundary) there; nets are numbered in DEF
    order.
Only the placement orientations N, S, FN, FS are supported (ispd18_test1 uses N and FS).
"""
from __future__ import annotations

import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from .regions import ACCESS, BLOCKAGE, NORMAL, Region, pack_records

Rect = Tuple[int, int, int, int]


@dataclass
class Macro:
    width: int = 0
    height: int = 0
    pins: Dict[str, dict] = field(default_factory=dict)      # name -> {"use": str, "rects": [(z, x0, y0, x1, y1)]}
    obs: List[Tuple[int, int, int, int, int]] = field(default_factory=list)


@dataclass
class Design:
    dbu: int = 2000
    die: Rect = (0, 0, 0, 0)
    layers: List[str] = field(default_factory=list)           # routing layers, bottom-up
    layer_dir: List[int] = field(default_factory=list)        # 0 horizontal (x moves), 1 vertical
    layer_bloat: List[int] = field(default_factory=list)      # DBU: width/2 + spacing (blockage shapes grow by this)
    tracks: Dict[int, Dict[str, Tuple[int, int, int]]] = field(default_factory=dict)   # z -> {"X": (start, n, step)}
    macros: Dict[str, Macro] = field(default_factory=dict)
    components: Dict[str, Tuple[str, int, int, str]] = field(default_factory=dict)
    nets: List[Tuple[str, List[Tuple[str, str]]]] = field(default_factory=list)
    guides: Dict[str, List[Tuple[int, int, int, int, int]]] = field(default_factory=dict)   # net -> [(x0,y0,x1,y1,z)]


def _tokens(path: str) -> List[str]:
    text = open(path).read()
    text = re.sub(r"#[^\n]*", "", text)
    return text.replace("(", " ( ").replace(")", " ) ").replace(";", " ; ").split()


def parse_lef(path: str, design: Optional[Design] = None) -> Design:
    d = design or Design()
    t = _tokens(path)
    i, n = 0, len(t)
    lef_dbu = 2000

    def to_dbu(v: str) -> int:
        return int(round(float(v) * lef_dbu))

    while i < n:
        tok = t[i]
        if tok == "UNITS":
            while t[i] != "END":
                if t[i] == "DATABASE" and t[i + 1] == "MICRONS":
                    lef_dbu = int(float(t[i + 2]))
                i += 1
            i += 2
        elif tok == "LAYER" and i + 2 < n and t[i + 2] != ";":
            name = t[i + 1]
            j = i + 2
            ltype, ldir, width, spacing = None, None, 0.0, None
            while not (t[j] == "END" and t[j + 1] == name):
                if t[j] == "TYPE":
                    ltype = t[j + 1]
                if t[j] == "DIRECTION":
                    ldir = t[j + 1]
                if t[j] == "WIDTH" and t[j + 2] == ";":
                    width = float(t[j + 1])
                if t[j] == "SPACING" and spacing is None:
                    try:
                        spacing = float(t[j + 1])
                    except ValueError:
                        pass
                j += 1
            if ltype == "ROUTING":
                d.layers.append(name)
                d.layer_dir.append(1 if ldir == "VERTICAL" else 0)
                d.layer_bloat.append(int(round((width / 2 + (spacing or 0.0)) * lef_dbu)))
            i = j + 2
        elif tok == "MACRO":
            name = t[i + 1]
            m = Macro()
            j = i + 2
            zmap = {nm: z for z, nm in enumerate(d.layers)}
            while not (t[j] == "END" and t[j + 1] == name):
                if t[j] == "SIZE":
                    m.width, m.height = to_dbu(t[j + 1]), to_dbu(t[j + 3])
                    j += 4
                elif t[j] == "PIN":
                    pname = t[j + 1]
                    pin = {"use": "SIGNAL", "rects": []}
                    j += 2
                    z = None
                    while not (t[j] == "END" and t[j + 1] == pname):
                        if t[j] == "USE":
                            pin["use"] = t[j + 1]
                        elif t[j] == "LAYER":
                            z = zmap.get(t[j + 1])
                        elif t[j] == "RECT" and z is not None:
                            pin["rects"].append((z, to_dbu(t[j + 1]), to_dbu(t[j + 2]), to_dbu(t[j + 3]), to_dbu(t[j + 4])))
                        j += 1
                    m.pins[pname] = pin
                    j += 2
                elif t[j] == "OBS":
                    j += 1
                    z = None
                    while t[j] != "END":
                        if t[j] == "LAYER":
                            z = zmap.get(t[j + 1])
                        elif t[j] == "RECT" and z is not None:
                            m.obs.append((z, to_dbu(t[j + 1]), to_dbu(t[j + 2]), to_dbu(t[j + 3]), to_dbu(t[j + 4])))
                        j += 1
                    j += 1
                else:
                    j += 1
            d.macros[name] = m
            i = j + 2
        else:
            i += 1
    d.dbu = lef_dbu
    return d


def parse_def(path: str, design: Design) -> Design:
    d = design
    t = _tokens(path)
    zmap = {nm: z for z, nm in enumerate(d.layers)}
    i, n = 0, len(t)
    while i < n:
        tok = t[i]
        if tok == "UNITS" and t[i + 1] == "DISTANCE":
            d.dbu = int(t[i + 3])
            i += 4
        elif tok == "DIEAREA":
            d.die = (int(t[i + 2]), int(t[i + 3]), int(t[i + 6]), int(t[i + 7]))
            i += 9
        elif tok == "TRACKS":
            axis, start, num, step = t[i + 1], int(t[i + 2]), int(t[i + 4]), int(t[i + 6])
            j = i + 7
            while t[j] != ";":
                if t[j] == "LAYER" and t[j + 1] in zmap:
                    d.tracks.setdefault(zmap[t[j + 1]], {})[axis] = (start, num, step)
                j += 1
            i = j + 1
        elif tok == "COMPONENTS":
            j = i + 3
            while not (t[j] == "END" and t[j + 1] == "COMPONENTS"):
                if t[j] == "-":
                    name, macro = t[j + 1], t[j + 2]
                    k = j + 3
                    x = y = 0
                    orient = "N"
                    while t[k] != ";":
                        if t[k] in ("PLACED", "FIXED", "COVER"):
                            x, y, orient = int(t[k + 2]), int(t[k + 3]), t[k + 5]
                        k += 1
                    d.components[name] = (macro, x, y, orient)
                    j = k
                j += 1
            i = j + 2
        elif tok == "NETS" and t[i - 1] != "END":
            j = i + 3
            while not (t[j] == "END" and t[j + 1] == "NETS"):
                if t[j] == "-":
                    name = t[j + 1]
                    conns = []
                    k = j + 2
                    while t[k] != ";":
                        if t[k] == "(" and t[k + 3] == ")":
                            conns.append((t[k + 1], t[k + 2]))
                            k += 4
                        else:
                            k += 1
                    d.nets.append((name, conns))
                    j = k
                j += 1
            i = j + 2
        else:
            i += 1
    return d


def parse_guide(path: str, design: Design) -> Design:
    zmap = {nm: z for z, nm in enumerate(design.layers)}
    net = None
    for line in open(path):
        s = line.split()
        if not s:
            continue
        if len(s) == 1 and s[0] not in ("(", ")"):
            net = s[0]
            design.guides[net] = []
        elif len(s) == 5 and net is not None and s[4] in zmap:
            design.guides[net].append((int(s[0]), int(s[1]), int(s[2]), int(s[3]), zmap[s[4]]))
    return design


def load_design(lef: str, deff: str, guide: Optional[str] = None) -> Design:
    d = parse_def(deff, parse_lef(lef))
    if guide:
        parse_guide(guide, d)
    return d


# ---------------------------------------------------------------------------------------------------------
def _place(rect, macro: Macro, px: int, py: int, orient: str) -> Rect:
    z, x0, y0, x1, y1 = rect
    w, h = macro.width, macro.height
    if orient == "N":
        r = (x0, y0, x1, y1)
    elif orient == "FS":
        r = (x0, h - y1, x1, h - y0)
    elif orient == "S":
        r = (w - x1, h - y1, w - x0, h - y0)
    elif orient == "FN":
        r = (w - x1, y0, w - x0, y1)
    else:
        raise NotImplementedError(f"placement orientation {orient}")
    return (px + r[0], py + r[1], px + r[2], py + r[3])


def _track_coords(design: Design, axis: str, lo: int, hi: int) -> np.ndarray:
    vals = []
    for z, tr in design.tracks.items():
        if axis in tr:
            start, num, step = tr[axis]
            k0 = max(0, -(-(lo - start) // step))
            k1 = min(num - 1, (hi - start) // step)
            if k1 >= k0:
                vals.append(start + step * np.arange(k0, k1 + 1, dtype=np.int64))
    return np.unique(np.concatenate(vals)) if vals else np.zeros(0, np.int64)


GUIDE_MAX_BOXES = 8        # XR_GUIDE_MAX_BOXES of include/xroute_hip.h


def merge_guide_boxes(boxes, limit: int = GUIDE_MAX_BOXES):
    """At most `limit` boxes (x0, y0, x1, y1, z0, z1) covering the given ones: duplicates dropped, equal footprints on adjacent
    layers joined into one layer range (a guide usually repeats a GCell footprint on the layers a via stack crosses); if that is
    still too many, one bounding box per layer range, then one box around everything.  Each step only ever GROWS the guide."""
    boxes = sorted(set(boxes), key=lambda g: (g[0], g[1], g[2], g[3], g[4]))
    out: List[Tuple[int, int, int, int, int, int]] = []
    for g in boxes:
        if out and out[-1][:4] == g[:4] and g[4] <= out[-1][5] + 1:
            out[-1] = out[-1][:5] + (max(out[-1][5], g[5]),)
        else:
            out.append(tuple(g))
    if len(out) <= limit:
        return out
    by_z: Dict[Tuple[int, int], List] = {}
    for g in out:
        by_z.setdefault((g[4], g[5]), []).append(g)
    out = [(min(g[0] for g in gs), min(g[1] for g in gs), max(g[2] for g in gs), max(g[3] for g in gs), z0, z1)
           for (z0, z1), gs in sorted(by_z.items())]
    if len(out) <= limit:
        return out
    return [(min(g[0] for g in out), min(g[1] for g in out), max(g[2] for g in out), max(g[3] for g in out),
             min(g[4] for g in out), max(g[5] for g in out))]


class RegionExtractor:
    """Spatial index over the placed design + per-box region extraction."""

    def __init__(self, design: Design, snap: int = 400):
        self.d = design
        self.snap = snap
        comp_net: Dict[Tuple[str, str], int] = {}
        for ni, (_, conns) in enumerate(design.nets):
            for c in conns:
                comp_net[c] = ni
        self.comp_net = comp_net
        names = list(design.components.keys())
        self.names = names
        boxes = np.zeros((len(names), 4), np.int64)
        for i, nm in enumerate(names):
            macro, x, y, _ = design.components[nm]
            m = design.macros[macro]
            boxes[i] = (x, y, x + m.width, y + m.height)
        self.boxes = boxes
        # guide rectangles as arrays per net index
        self.net_index = {nm: i for i, (nm, _) in enumerate(design.nets)}

    def extract(self, box: Rect, name: str = "", route_box: Optional[Rect] = None, drc_halo: int = 500) -> Region:
        """The region of `box` (the worker's extBox: every track inside it is routing resource).  `route_box` (the worker's routeBox,
        inside `box`) restricts which nets are ROUTED here, the way the reference describes its one static region (xroute_env/__init__.py:
        13-23: a 1x1-GCell routeBox of ispd18_test1 with 36 nets): a net is routed when its global-route guide overlaps the routeBox or
        one of its cell pins lies in routeBox + drcBox (`drc_halo`, 500 DBU in the shipped worker dumps); every other net
        met in the halo is an obstacle (its pin shapes are blocked), the halo's free tracks stay routing resource.  With a routeBox the
        worker only routes what lies inside it: a cell pin outside routeBox + drcBox is not a pin of the region (its net's wiring out
        there exists already: the shape is an obstacle), and a net that leaves the routeBox gets its boundary pin where its guide crosses
        the ROUTEBOX edge, not the extBox edge.  None: every net with two pins inside `box`, boundary pins on the edge of `box`
        (round 2-3 behaviour)."""
        d = self.d
        bx0, by0, bx1, by1 = box
        selected = None
        pbx0, pby0, pbx1, pby1 = box                       # the box pins live in / boundary pins sit on the edge of
        if route_box is not None:
            rx0, ry0, rx1, ry1 = route_box
            pbx0, pby0, pbx1, pby1 = route_box
            selected = set()
            for nm, rects in d.guides.items():
                ni = self.net_index.get(nm)
                if ni is None:
                    continue
                for (gx0, gy0, gx1, gy1, z) in rects:
                    if max(gx0, rx0) < min(gx1, rx1) and max(gy0, ry0) < min(gy1, ry1):
                        selected.add(ni)
                        break
        xs = _track_coords(d, "X", bx0, bx1)
        ys = _track_coords(d, "Y", by0, by1)
        Z = len(d.layers)
        X, Y = len(xs), len(ys)
        if X == 0 or Y == 0:
            raise ValueError("no tracks inside the box")
        ntype = np.full((X, Y, Z), NORMAL, np.int64)
        owner_net = np.full((X, Y, Z), -1, np.int64)     # design net index of an ACCESS node
        owner_pin = np.full((X, Y, Z), -1, np.int64)     # pin slot inside that net (design-wide conn index)

        def span(c, lo, hi):
            return int(np.searchsorted(c, lo, "left")), int(np.searchsorted(c, hi, "right"))

        hit = np.flatnonzero((self.boxes[:, 0] <= bx1) & (self.boxes[:, 2] >= bx0) &
                             (self.boxes[:, 1] <= by1) & (self.boxes[:, 3] >= by0))
        pin_nodes: Dict[Tuple[int, int], List[Tuple[int, int, int]]] = {}   # (net, conn idx) -> nodes
        pin_rect_center: Dict[Tuple[int, int], Tuple[int, int, int]] = {}
        for ci in hit:
            cname = self.names[ci]
            macro_name, px, py, orient = d.components[cname]
            m = d.macros[macro_name]
            for rect in m.obs:
                x0, y0, x1, y1 = _place(rect, m, px, py, orient)
                g = d.layer_bloat[rect[0]]
                a, b = span(xs, x0 - g, x1 + g); c, e = span(ys, y0 - g, y1 + g)
                if b > a and e > c:
                    ntype[a:b, c:e, rect[0]] = BLOCKAGE
            for pname, pin in m.pins.items():
                ni = self.comp_net.get((cname, pname), -1)
                routed = pin["use"] == "SIGNAL" and ni >= 0
                if routed and route_box is not None:       # a pin is the region's when (the centre of) one of its shapes lies in routeBox + drcBox
                    routed = False
                    for rect in pin["rects"]:
                        x0, y0, x1, y1 = _place(rect, m, px, py, orient)
                        if (rx0 - drc_halo <= (x0 + x1) // 2 <= rx1 + drc_halo and ry0 - drc_halo <= (y0 + y1) // 2 <= ry1 + drc_halo):
                            routed = True
                            break
                key = None
                if routed:
                    conns = d.nets[ni][1]
                    key = (ni, conns.index((cname, pname)))
                for rect in pin["rects"]:
                    x0, y0, x1, y1 = _place(rect, m, px, py, orient)
                    a, b = span(xs, x0, x1); c, e = span(ys, y0, y1)
                    z = rect[0]
                    if routed:
                        if selected is not None and (rx0 - drc_halo <= (x0 + x1) // 2 <= rx1 + drc_halo and
                                                     ry0 - drc_halo <= (y0 + y1) // 2 <= ry1 + drc_halo):
                            selected.add(ni)
                        if key not in pin_rect_center and bx0 <= (x0 + x1) // 2 <= bx1 and by0 <= (y0 + y1) // 2 <= by1:
                            pin_rect_center[key] = ((x0 + x1) // 2, (y0 + y1) // 2, z)
                        for xi in range(a, b):
                            for yj in range(c, e):
                                pin_nodes.setdefault(key, []).append((xi, yj, z))
                    else:
                        g = d.layer_bloat[z]
                        a, b = span(xs, x0 - g, x1 + g); c, e = span(ys, y0 - g, y1 + g)
                        if b > a and e > c:
                            ntype[a:b, c:e, z] = BLOCKAGE
        # pins without an on-grid node: snap to the nearest node of the pin's layer
        for key, (cx, cy, z) in pin_rect_center.items():
            if key in pin_nodes:
                continue
            xi = int(np.argmin(np.abs(xs - cx))); yj = int(np.argmin(np.abs(ys - cy)))
            if abs(int(xs[xi]) - cx) <= self.snap and abs(int(ys[yj]) - cy) <= self.snap:
                pin_nodes[key] = [(xi, yj, z)]
        # boundary pins from the global-route guides, on the edge of the pin box (the routeBox when there is one)
        xi_lo, xi_hi = int(np.searchsorted(xs, pbx0, "left")), int(np.searchsorted(xs, pbx1, "right")) - 1
        yj_lo, yj_hi = int(np.searchsorted(ys, pby0, "left")), int(np.searchsorted(ys, pby1, "right")) - 1
        xi_lo, xi_hi, yj_lo, yj_hi = max(0, min(xi_lo, X - 1)), max(0, min(xi_hi, X - 1)), max(0, min(yj_lo, Y - 1)), max(0, min(yj_hi, Y - 1))
        boundary: Dict[int, List[Tuple[int, int, int]]] = {}
        for nm, rects in d.guides.items():
            ni = self.net_index.get(nm)
            if ni is None:
                continue
            for (gx0, gy0, gx1, gy1, z) in rects:
                cx0, cy0, cx1, cy1 = max(gx0, pbx0), max(gy0, pby0), min(gx1, pbx1), min(gy1, pby1)
                if cx0 >= cx1 or cy0 >= cy1:
                    continue
                vert = d.layer_dir[z] == 1
                if not vert:       # horizontal wires leave through the left / right edge
                    yj = int(np.argmin(np.abs(ys - (cy0 + cy1) // 2)))
                    if gx0 < pbx0:
                        boundary.setdefault(ni, []).append((xi_lo, yj, z))
                    if gx1 > pbx1:
                        boundary.setdefault(ni, []).append((xi_hi, yj, z))
                else:
                    xi = int(np.argmin(np.abs(xs - (cx0 + cx1) // 2)))
                    if gy0 < pby0:
                        boundary.setdefault(ni, []).append((xi, yj_lo, z))
                    if gy1 > pby1:
                        boundary.setdefault(ni, []).append((xi, yj_hi, z))
        # candidate pins per net: local pins first (DEF connection order), then boundary pins
        pins_of: Dict[int, List[List[Tuple[int, int, int]]]] = {}
        for (ni, _), nodes in sorted(pin_nodes.items()):
            pins_of.setdefault(ni, []).append(nodes)
        for ni, nodes in boundary.items():
            for nd in dict.fromkeys(nodes):
                pins_of.setdefault(ni, []).append([nd])
        # claim nodes: a node belongs to the first pin that asks for it; a boundary pin whose node is taken slides
        # along its edge to the nearest free node (up to 3 tracks)
        claimed: Dict[Tuple[int, int, int], Tuple[int, int]] = {}
        placed: Dict[int, List[List[Tuple[int, int, int]]]] = {}
        for ni in sorted(pins_of):
            for pi, nodes in enumerate(pins_of[ni]):
                got = [nd for nd in nodes if nd not in claimed]
                if not got and len(nodes) == 1:
                    xi, yj, z = nodes[0]
                    on_x_edge = xi in (xi_lo, xi_hi)
                    for dlt in (1, -1, 2, -2, 3, -3):
                        cand = (xi, yj + dlt, z) if on_x_edge else (xi + dlt, yj, z)
                        if 0 <= cand[0] < X and 0 <= cand[1] < Y and cand not in claimed:
                            got = [cand]
                            break
                for nd in got:
                    claimed[nd] = (ni, pi)
                if got:
                    placed.setdefault(ni, []).append(got)
        # a net is routed here when at least two of its pins got nodes (and the routeBox rule selects it); everything else is an obstacle
        routed_nets = sorted(ni for ni, pins in placed.items() if len(pins) >= 2 and (selected is None or ni in selected))
        local_id = {ni: k for k, ni in enumerate(routed_nets)}
        net = np.full((X, Y, Z), -1, np.int64)
        pin = np.full((X, Y, Z), -1, np.int64)
        for ni, pins in placed.items():
            for pi, nodes in enumerate(pins):
                for (xi, yj, z) in nodes:
                    if ni in local_id:
                        ntype[xi, yj, z] = ACCESS
                        net[xi, yj, z] = local_id[ni]
                        pin[xi, yj, z] = pi
                    elif ntype[xi, yj, z] == NORMAL:
                        ntype[xi, yj, z] = BLOCKAGE
        used = (ntype == BLOCKAGE).astype(np.int64)
        rec = pack_records(ntype.reshape(-1), used.reshape(-1), net.reshape(-1), pin.reshape(-1))
        # the routed nets' global-route guides inside the box, as boxes of track / layer indices (XR-Maze v2 `guide_cost`)
        guide_off = np.zeros(len(routed_nets) + 1, np.int32)
        guide_box: List[Tuple[int, int, int, int, int, int]] = []
        if d.guides:
            names = [d.nets[ni][0] for ni in routed_nets]
            for k, nm in enumerate(names):
                boxes = []
                for (gx0, gy0, gx1, gy1, z) in d.guides.get(nm, ()):
                    a, b = span(xs, max(gx0, bx0), min(gx1, bx1)); c, e = span(ys, max(gy0, by0), min(gy1, by1))
                    if b > a and e > c:
                        boxes.append((a, c, b - 1, e - 1, z, z))
                guide_box += merge_guide_boxes(boxes)
                guide_off[k + 1] = len(guide_box)
        reg = Region((X, Y, Z), xs.astype(np.int32), ys.astype(np.int32), np.asarray(d.layer_dir, np.uint8), rec,
                     len(routed_nets), np.zeros(3, np.int32), name or f"box{bx0}_{by0}")
        if d.guides:
            reg.guide_off = guide_off
            reg.guide_box = np.asarray(guide_box, np.int16).reshape(-1, 6)
        return reg

    def gcell_regions(self, gcell=(6000, 5700), halo: int = 2000, limit: Optional[int] = None, min_nets: int = 1,
                      route_box_rule: bool = True):
        """Regions of 1x1 GCell + halo over the die, row-major, skipping boxes with fewer than `min_nets` routed nets.  The GCell is the
        routeBox, the halo the extBox ring (routeBox 5700^2 + 2000 DBU in the reference's worker dumps): with `route_box_rule` only nets
        the routeBox rule of `extract` selects are routed; the halo is routing resource."""
        x0, y0, x1, y1 = self.d.die
        out = []
        for gy in range(0, (y1 - y0 + gcell[1] - 1) // gcell[1]):
            for gx in range(0, (x1 - x0 + gcell[0] - 1) // gcell[0]):
                rb = (x0 + gx * gcell[0], y0 + gy * gcell[1], min(x1, x0 + (gx + 1) * gcell[0]), min(y1, y0 + (gy + 1) * gcell[1]))
                box = (max(x0, rb[0] - halo), max(y0, rb[1] - halo), min(x1, rb[2] + halo), min(y1, rb[3] + halo))
                reg = self.extract(box, name=f"gcell_x{gx}_y{gy}", route_box=rb if route_box_rule else None)
                if reg.n_nets >= min_nets:
                    out.append(reg)
                    if limit is not None and len(out) >= limit:
                        return out
        return out


# ---------------------------------------------------------------------------------------------------------
# compact on-disk pack of regions (sparse: only non-NORMAL nodes are stored)
# ---------------------------------------------------------------------------------------------------------
def save_region_pack(path: str, regions: List[Region]):
    out = {"n": np.array(len(regions))}
    for i, r in enumerate(regions):
        idx = np.flatnonzero(r.nodes != np.uint32(NORMAL)).astype(np.int32)
        out[f"r{i}_dims"] = np.array(r.dims, np.int32)
        out[f"r{i}_xs"], out[f"r{i}_ys"] = r.xs, r.ys
        out[f"r{i}_ldir"] = r.layer_dir
        out[f"r{i}_idx"], out[f"r{i}_rec"] = idx, r.nodes[idx]
        out[f"r{i}_k"] = np.array(r.n_nets)
        if r.name:
            out[f"r{i}_name"] = np.array(str(r.name))
        if r.guide_off is not None:
            out[f"r{i}_goff"] = np.asarray(r.guide_off, np.int32)
            out[f"r{i}_gbox"] = np.asarray(r.guide_box, np.int16).reshape(-1, 6)
    np.savez_compressed(path, **out)


def load_region_pack(path: str) -> List[Region]:
    z = np.load(path)
    regs = []
    for i in range(int(z["n"])):
        dims = tuple(int(v) for v in z[f"r{i}_dims"])
        nodes = np.full(dims[0] * dims[1] * dims[2], NORMAL, np.uint32)
        nodes[z[f"r{i}_idx"]] = z[f"r{i}_rec"]
        regs.append(Region(dims, z[f"r{i}_xs"], z[f"r{i}_ys"], z[f"r{i}_ldir"], nodes, int(z[f"r{i}_k"]),
                           np.zeros(3, np.int32), str(z[f"r{i}_name"]) if f"r{i}_name" in z.files else f"pack{i}"))
        if f"r{i}_goff" in z.files:
            regs[-1].guide_off, regs[-1].guide_box = z[f"r{i}_goff"], z[f"r{i}_gbox"]
    return regs
