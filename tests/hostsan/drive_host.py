"""Driver of tests/test_host_sanitizers.py: runs in a child process with the sanitizer runtime preloaded (argv: repo root, library)."""
import ctypes as C, importlib.util, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from xroute_env_amd import _lib as X                      # ctypes structures only: the module does not load the real library
from xroute_env_amd.regions import generate_region, pack_records, BLOCKAGE, NORMAL, ACCESS
L = C.CDLL(sys.argv[2])
vp = C.c_void_p
L.xr_last_error.restype = C.c_char_p
L.xr_config_default.argtypes = [C.POINTER(X.XrConfig)]; L.xr_config_default.restype = None
L.xr_batch_create.argtypes = [C.POINTER(X.XrConfig), C.POINTER(vp)]
L.xr_batch_destroy.argtypes = [vp]
L.xr_batch_load_regions.argtypes = [vp, C.POINTER(X.XrRegionDesc), C.c_int32, vp]
L.xr_batch_assign.argtypes = [vp, vp]
L.xr_batch_sizes.argtypes = [vp] + [C.POINTER(C.c_int32)] * 6 + [C.POINTER(C.c_int64)]
L.xr_batch_reset.argtypes = [vp, vp, C.c_int32, vp]
L.xr_batch_step.argtypes = [vp, vp, vp]
L.xr_batch_step_observe.argtypes = [vp, vp, vp, C.c_int64, vp]
L.xr_batch_step_observe_inplace.argtypes = [vp, vp, vp, C.c_int64, vp]
L.xr_batch_step_compact.argtypes = [vp, vp, vp, C.c_int64, vp]
L.xr_batch_net_planes.argtypes = [vp, vp, vp, C.c_int32, vp, C.c_int64, vp]
L.xr_batch_random_actions.argtypes = [vp, vp, C.c_uint64, vp]
L.xr_batch_observation.argtypes = [vp, vp, C.c_int64, C.c_int32, C.c_int32, vp]
L.xr_batch_fetch.argtypes = [vp, C.c_int32, vp, C.c_size_t, vp]
L.xr_batch_store.argtypes = [vp, C.c_int32, vp, C.c_size_t, vp]
L.xr_batch_load_guides.argtypes = [vp, vp, vp, vp]
L.xr_batch_state_row_bytes.argtypes = [vp, C.POINTER(C.c_int64)]
L.xr_batch_pack_state.argtypes = [vp, vp, C.c_int64, C.c_int32, vp]
L.xr_batch_expand_state.argtypes = [vp, vp, C.c_int64, C.c_int32, vp, C.c_int64, vp, vp, vp]
L.xr_batch_route_order.argtypes = [vp, vp, C.c_int32, vp, vp]
L.xr_batch_ingest_state.argtypes = [vp, vp, vp, vp, vp]
L.xr_batch_route_occupancy.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
L.xr_batch_observe_timing.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_float)]
L.xr_observation_from_records.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, vp, C.c_int32, vp, vp]
limit = C.c_int64.in_dll(L, "xr_stub_alloc_limit"); live = C.c_int64.in_dll(L, "xr_stub_alloc_live"); launches = C.c_int64.in_dll(L, "xr_stub_launches")
rng = np.random.default_rng(99)
counts = {}
def note(rc): counts[rc] = counts.get(rc, 0) + 1; return rc

def cfg(**kw):
    c = X.XrConfig(); L.xr_config_default(C.byref(c))
    for k, v in kw.items(): setattr(c, k, v)
    return c

def create(**kw):
    h = vp(); rc = L.xr_batch_create(C.byref(cfg(**kw)), C.byref(h))
    return rc, h

keep = []
def desc(reg, **over):
    d = X.XrRegionDesc()
    xs = np.ascontiguousarray(over.get("xs", reg.xs), np.int32); ys = np.ascontiguousarray(over.get("ys", reg.ys), np.int32)
    ld = np.ascontiguousarray(over.get("layer_dir", reg.layer_dir), np.uint8); nodes = np.ascontiguousarray(over.get("nodes", reg.nodes), np.uint32)
    keep.extend([xs, ys, ld, nodes])
    d.dim_x, d.dim_y, d.dim_z = over.get("dims", reg.dims)
    d.xs_host, d.ys_host, d.layer_dir_host, d.nodes_host = xs.ctypes.data, ys.ctypes.data, ld.ctypes.data, nodes.ctypes.data
    for k in ("xs_host", "ys_host", "layer_dir_host", "nodes_host"):
        if k in over: setattr(d, k, over[k])
    d.n_nets = over.get("n_nets", reg.n_nets)
    m0 = over.get("metrics0", reg.metrics0)
    d.metrics0[0], d.metrics0[1], d.metrics0[2] = int(m0[0]), int(m0[1]), int(m0[2])
    return d

def load(h, descs):
    arr = (X.XrRegionDesc * len(descs))(*descs)
    return L.xr_batch_load_regions(h, arr, len(descs), None)

# ---- 1. xr_batch_create: every field out of range
bad_cfgs = [dict(struct_size=4), dict(n_envs=0), dict(n_envs=-5), dict(via_cost=0), dict(drc_cost=-1), dict(drc_unit=-1), dict(max_route_count=0),
            dict(drc_cost=1 << 20, drc_unit=1 << 20), dict(via_cost=1 << 22), dict(launch_order=3), dict(launch_order=-1), dict(obs_mode=4), dict(obs_mode=-1),
            dict(obs_writer_blocks=-1), dict(obs_split_permille=1001), dict(router=4), dict(router=-1), dict(dial_mult=65), dict(dial_mult=-1),
            dict(guide_cost=-1), dict(guide_cost=1 << 22), dict(guide_margin=-1), dict(maze_end_iter=0), dict(maze_end_iter=9), dict(maze_end_iter=8, drc_cost=8, drc_unit=40000),
            dict(stream_per_region=2), dict(stream_per_region=1, n_envs=65), dict(debug_round_cap=-1), dict(obs_helper_blocks=-1), dict(block_threads=63),
            dict(block_threads=2048), dict(block_threads=96), dict(device=1), dict(device=-1)]
for kw in bad_cfgs:
    rc, h = create(**kw)
    assert rc < 0 and not h.value, (kw, rc)
    assert len(L.xr_last_error()) > 0
assert L.xr_batch_create(None, None) == X.XR_ERR_INVALID
assert L.xr_batch_destroy(None) == 0

# ---- 2. every entry point before load_regions: XR_ERR_STATE (or INVALID for null), never a crash
rc, h = create(n_envs=3); assert rc == 0
buf = np.zeros(1 << 16, np.uint8); p = buf.ctypes.data
i32 = C.c_int32(); i64 = C.c_int64(); f32 = C.c_float()
pre = [L.xr_batch_assign(h, p), L.xr_batch_sizes(h, None, None, None, None, None, None, None), L.xr_batch_reset(h, None, 0, None), L.xr_batch_step(h, p, None),
       L.xr_batch_step_observe(h, p, p, 1 << 20, None), L.xr_batch_step_observe_inplace(h, p, p, 1 << 20, None), L.xr_batch_step_compact(h, p, p, 1 << 20, None),
       L.xr_batch_net_planes(h, p, p, 1, p, 1 << 20, None), L.xr_batch_random_actions(h, p, 1, None), L.xr_batch_observation(h, p, 1 << 20, 0, 1, None),
       L.xr_batch_fetch(h, 0, p, 1 << 16, None), L.xr_batch_store(h, 0, p, 36, None), L.xr_batch_load_guides(h, None, None, None),
       L.xr_batch_state_row_bytes(h, C.byref(i64)), L.xr_batch_pack_state(h, p, 1 << 10, 0, None), L.xr_batch_expand_state(h, p, 1 << 10, 1, p, 1 << 20, p, p, None),
       L.xr_batch_route_order(h, p, 100, None, None), L.xr_batch_route_occupancy(h, C.byref(i32), C.byref(i64)), L.xr_batch_ingest_state(h, p, p, p, None)]
assert all(r == X.XR_ERR_STATE for r in pre), pre
for fn, a in ((L.xr_batch_assign, (None, None)), (L.xr_batch_step, (None, None, None)), (L.xr_batch_fetch, (None, 0, None, 0, None)), (L.xr_batch_store, (None, 0, None, 0, None)), (L.xr_batch_state_row_bytes, (None, None))):
    assert fn(*a) == X.XR_ERR_INVALID
L.xr_batch_destroy(h)

# ---- 3. valid loads of many shapes and configs: every staging path, every router placement
shapes = [((1, 1, 1), (0, 1)), ((2, 1, 1), (1, 1)), ((3, 9, 2), (1, 4)), ((7, 5, 3), (2, 6)), ((24, 40, 9), (4, 12)), ((25, 34, 9), (1, 30)), ((6, 6, 12), (1, 5)),
          ((40, 40, 12), (2, 8)), ((5, 4, 32), (1, 3))]
cfgs = [dict(n_envs=1), dict(n_envs=5), dict(n_envs=70, router=1), dict(n_envs=4, router=2), dict(n_envs=4, router=3), dict(n_envs=2, force_scratch_field=1),
        dict(n_envs=3, guide_cost=500, guide_margin=1, maze_end_iter=3), dict(n_envs=8, stream_per_region=1), dict(n_envs=2, obs_mode=1), dict(n_envs=2, obs_mode=2, obs_split_permille=300),
        dict(n_envs=2, path_cap=7), dict(n_envs=2, window=20, force_scratch_field=1), dict(n_envs=4200, auto_reset=1)]
n_loaded = 0
for ci, kw in enumerate(cfgs):
    rc, h = create(**kw); assert rc == 0, kw
    B = kw["n_envs"]
    for si in range(3):
        pick = [shapes[(ci + si + j) % len(shapes)] for j in range(1 + (ci + si) % 3)]
        regs = [generate_region(4000 + 10 * ci + si + j, dims=d, k_range=k, net_span=4) for j, (d, k) in enumerate(pick)]
        rc = load(h, [desc(r) for r in regs])
        note(rc)
        if rc != 0:
            assert rc in (X.XR_ERR_RANGE,), (kw, pick, rc, L.xr_last_error())
            continue
        n_loaded += 1
        ne, nr, nm, km, lw, pc = (C.c_int32() for _ in range(6)); st = C.c_int64()
        assert L.xr_batch_sizes(h, C.byref(ne), C.byref(nr), C.byref(nm), C.byref(km), C.byref(lw), C.byref(pc), C.byref(st)) == 0
        assert ne.value == B and nr.value == len(regs) and nm.value >= max(r.n_nodes for r in regs)
        N, K = nm.value, km.value
        act = np.ones(B, np.int32); out = np.zeros(8, np.float32)
        big = st.value
        # strides too small / misaligned / null
        need = (2 + 7 * K) * max(r.n_nodes for r in regs)
        assert L.xr_batch_step_observe(h, act.ctypes.data, out.ctypes.data, need - 1, None) == X.XR_ERR_RANGE
        assert L.xr_batch_step_observe(h, None, out.ctypes.data, big, None) == X.XR_ERR_INVALID
        assert L.xr_batch_step_compact(h, act.ctypes.data, out.ctypes.data, 2 * max(r.n_nodes for r in regs) - 1, None) == X.XR_ERR_RANGE
        assert L.xr_batch_observation(h, out.ctypes.data, big, 0, B + 1, None) == X.XR_ERR_RANGE
        assert L.xr_batch_observation(h, out.ctypes.data, big, 2, 1, None) == X.XR_ERR_RANGE
        assert L.xr_batch_observation(h, out.ctypes.data, 1, 0, B, None) == X.XR_ERR_RANGE
        assert L.xr_batch_net_planes(h, act.ctypes.data, act.ctypes.data, -1, out.ctypes.data, 7 * N, None) == X.XR_ERR_RANGE
        assert L.xr_batch_net_planes(h, act.ctypes.data, act.ctypes.data, 1, out.ctypes.data, 7 * max(r.n_nodes for r in regs) - 1, None) == X.XR_ERR_RANGE
        assert L.xr_batch_route_order(h, act.ctypes.data, K - 1, None, None) == X.XR_ERR_RANGE
        # the launches themselves (no-ops here) with every form's host-side bookkeeping
        l0 = launches.value
        for fn in (L.xr_batch_step_observe, L.xr_batch_step_observe_inplace, L.xr_batch_step_observe_inplace):
            assert fn(h, act.ctypes.data, out.ctypes.data, big, None) == 0
        assert L.xr_batch_step(h, act.ctypes.data, None) == 0
        rc = L.xr_batch_step_compact(h, act.ctypes.data, out.ctypes.data, (2 * N + 3) & ~3, None); assert rc in (0, X.XR_ERR_INVALID), rc
        assert L.xr_batch_reset(h, None, 1, None) == 0 and L.xr_batch_random_actions(h, act.ctypes.data, 5, None) == 0
        assert L.xr_batch_observation(h, out.ctypes.data, big, 0, B, None) == 0
        assert L.xr_batch_route_occupancy(h, C.byref(i32), C.byref(i64)) == 0 and L.xr_batch_observe_timing(h, C.byref(i32), C.byref(f32)) == 0
        assert launches.value > l0
        # assign
        bad = np.zeros(B, np.int32); bad[-1] = len(regs)
        assert L.xr_batch_assign(h, bad.ctypes.data) == X.XR_ERR_RANGE
        bad[-1] = -1; assert L.xr_batch_assign(h, bad.ctypes.data) == X.XR_ERR_RANGE
        good = (np.arange(B) % len(regs)).astype(np.int32); assert L.xr_batch_assign(h, good.ctypes.data) == 0
        # fetch: every selector into an exact-size buffer (a red zone right behind it), one byte short is refused
        lwv, pcv = lw.value, pc.value
        sizes = {X.XR_FETCH_CUM: 12 * B, X.XR_FETCH_DELTA: 12 * B, X.XR_FETCH_REWARD: 8 * B, X.XR_FETCH_DONE: B, X.XR_FETCH_NLEGAL: 4 * B, X.XR_FETCH_STATUS: 4 * B,
                 X.XR_FETCH_LEGAL: 8 * B * lwv, X.XR_FETCH_PATH_LEN: 4 * B, X.XR_FETCH_PATH: 4 * B * pcv, X.XR_FETCH_OWNER: 2 * B * N, X.XR_FETCH_HASH: 8 * B,
                 X.XR_FETCH_REGION: 4 * B, X.XR_FETCH_STEPS: 8, X.XR_FETCH_SWEEPS: 4 * B, X.XR_FETCH_PHASES: 64 * B, X.XR_FETCH_RECORD: 48 * B, X.XR_FETCH_TOUCHED: 4 * B,
                 X.XR_FETCH_UNITS: 4, X.XR_FETCH_ROUTE_ORDER: 4 * B, X.XR_FETCH_REPLAY: 4 * B, X.XR_FETCH_ENV_STEPS: 8 * B}
        assert len(sizes) == 21
        for what in (-1, 21, 22, 1000):
            assert L.xr_batch_fetch(h, what, p, 1 << 16, None) == X.XR_ERR_INVALID
        for what, nb in sizes.items():
            exact = np.zeros(nb, np.uint8)
            assert L.xr_batch_fetch(h, what, exact.ctypes.data, nb, None) == 0, (what, nb, L.xr_last_error())
            assert L.xr_batch_fetch(h, what, exact.ctypes.data, nb - 1, None) == X.XR_ERR_RANGE
        # store: wrong sizes, hostile contents for the arrays kernels index with
        for what, nb in sizes.items():
            src = rng.integers(0, 256, max(nb, 1), dtype=np.uint8)
            rc = L.xr_batch_store(h, what, src.ctypes.data, nb, None)
            assert rc in (0, X.XR_ERR_RANGE, X.XR_ERR_INVALID), (what, rc)
            note(rc)
            assert L.xr_batch_store(h, what, src.ctypes.data, nb + 1, None) in (X.XR_ERR_RANGE, X.XR_ERR_INVALID)
        assert L.xr_batch_assign(h, good.ctypes.data) == 0
        r_bad = good.copy(); r_bad[0] = len(regs) + 3
        assert L.xr_batch_store(h, X.XR_FETCH_REGION, r_bad.ctypes.data, 4 * B, None) == X.XR_ERR_RANGE
        nl = np.array([regs[g].n_nets for g in good], np.int32); assert L.xr_batch_store(h, X.XR_FETCH_NLEGAL, nl.ctypes.data, 4 * B, None) == 0
        nl[0] += 1; assert L.xr_batch_store(h, X.XR_FETCH_NLEGAL, nl.ctypes.data, 4 * B, None) == X.XR_ERR_RANGE
        lg = np.zeros((B, lw.value), np.uint64)
        for e in range(B):
            for n in range(regs[good[e]].n_nets): lg[e, n >> 6] |= np.uint64(1) << np.uint64(n & 63)
        assert L.xr_batch_store(h, X.XR_FETCH_LEGAL, lg.ctypes.data, lg.nbytes, None) == 0
        kk = regs[good[0]].n_nets
        if kk < 64 * lw.value:
            lg[0, kk >> 6] |= np.uint64(1) << np.uint64(kk & 63)
            assert L.xr_batch_store(h, X.XR_FETCH_LEGAL, lg.ctypes.data, lg.nbytes, None) == X.XR_ERR_RANGE
        # state rows
        rb = C.c_int64(); assert L.xr_batch_state_row_bytes(h, C.byref(rb)) == 0
        rows = np.zeros(B * rb.value + 16, np.uint8); rp = rows.ctypes.data; rp += (-rp) % 8
        assert L.xr_batch_pack_state(h, rp, rb.value, 0, None) == 0
        assert L.xr_batch_pack_state(h, rp, rb.value - 8, 0, None) == X.XR_ERR_RANGE and L.xr_batch_pack_state(h, rp + 4, rb.value, 0, None) == X.XR_ERR_RANGE
        assert L.xr_batch_pack_state(h, rp, rb.value + 4, 0, None) == X.XR_ERR_RANGE and L.xr_batch_pack_state(h, rp, rb.value, -1, None) == X.XR_ERR_RANGE
        assert L.xr_batch_expand_state(h, rp, rb.value, B, out.ctypes.data, 2 * N, act.ctypes.data, act.ctypes.data, None) == 0
        assert L.xr_batch_expand_state(h, rp, 8, B, out.ctypes.data, 2 * N, act.ctypes.data, act.ctypes.data, None) == X.XR_ERR_RANGE
        assert L.xr_batch_expand_state(h, rp, rb.value, -1, out.ctypes.data, 2 * N, act.ctypes.data, act.ctypes.data, None) == X.XR_ERR_RANGE
        assert L.xr_batch_expand_state(h, rp, rb.value, B, None, 2 * N, act.ctypes.data, act.ctypes.data, None) == X.XR_ERR_INVALID
        assert L.xr_batch_expand_state(h, rp, rb.value, B, out.ctypes.data, 1, act.ctypes.data, act.ctypes.data, None) == X.XR_ERR_RANGE
        # ingest of an external state: null / misaligned pointers are refused
        ow = np.zeros(B * N + 8, np.int16); op = ow.ctypes.data; op += (-op) % 16
        lgw = np.zeros(B * lwv + 1, np.uint64); cm = np.zeros(B * 3, np.int32)
        assert L.xr_batch_ingest_state(h, op, lgw.ctypes.data, cm.ctypes.data, None) == 0
        assert L.xr_batch_ingest_state(h, op + 2, lgw.ctypes.data, cm.ctypes.data, None) == X.XR_ERR_INVALID
        assert L.xr_batch_ingest_state(h, op, lgw.ctypes.data + 4, cm.ctypes.data, None) == X.XR_ERR_INVALID
        assert L.xr_batch_ingest_state(h, op, None, cm.ctypes.data, None) == X.XR_ERR_INVALID
        # guides: valid tables, then hostile ones; a refused table leaves the batch usable
        offs, boxes = [], []
        for r in regs:
            nb = rng.integers(0, 3, r.n_nets)
            o = np.concatenate([[0], np.cumsum(nb)]).astype(np.int32)
            bx = np.zeros((int(o[-1]), 6), np.int16)
            for i in range(len(bx)):
                x0, x1 = sorted(rng.integers(0, r.dims[0], 2)); y0, y1 = sorted(rng.integers(0, r.dims[1], 2)); z0, z1 = sorted(rng.integers(0, r.dims[2], 2))
                bx[i] = (x0, y0, x1, y1, z0, z1)
            offs.append(o); boxes.append(bx)
        def guides(offs, boxes, null_off=(), null_box=()):
            oa = (vp * len(offs))(*[None if i in null_off else o.ctypes.data for i, o in enumerate(offs)])
            ba = (vp * len(boxes))(*[None if i in null_box else (b_.ctypes.data if b_.size else None) for i, b_ in enumerate(boxes)])
            return L.xr_batch_load_guides(h, oa, ba, None)
        assert guides(offs, boxes) == 0
        assert guides(offs, boxes, null_off=(0,)) == 0
        assert L.xr_batch_load_guides(h, None, None, None) == 0
        for trial in range(12):
            o2 = [o.copy() for o in offs]; b2 = [b_.copy() for b_ in boxes]
            ri = int(rng.integers(len(regs)))
            kind = trial % 6
            if regs[ri].n_nets == 0: continue
            ni = int(rng.integers(regs[ri].n_nets))
            if kind == 0: o2[ri][ni + 1:] -= 5 if o2[ri][ni + 1] < 5 else o2[ri][ni + 1] + 1           # descending / negative offsets
            elif kind == 1: o2[ri][ni + 1:] += 17                                                        # more than the box cap for one net
            elif kind == 2 and b2[ri].size: b2[ri][int(rng.integers(len(b2[ri]))), int(rng.integers(6))] = int(rng.choice([-1, 32767, -32768, 200]))
            elif kind == 3 and b2[ri].size: b2[ri][int(rng.integers(len(b2[ri])))] = (3, 0, 1, 0, 0, 0)  # x1 < x0
            elif kind == 4: o2[ri][0] = -1
            else: o2[ri][-1] = o2[ri][-1] + 1                                                              # reads one box past the table unless refused
            if kind == 1:
                b2[ri] = np.zeros((int(o2[ri][-1]) + 1, 6), np.int16)
            if kind == 5:
                b2[ri] = np.concatenate([b2[ri], np.full((1, 6), 30000, np.int16)])
            rc = guides(o2, b2)
            assert rc in (0, X.XR_ERR_RANGE), rc
            note(rc)
            assert L.xr_batch_step(h, act.ctypes.data, None) == 0
    L.xr_batch_destroy(h)
    assert live.value == 0, ("leaked device bytes", live.value, kw)
assert n_loaded >= 25, n_loaded

# ---- 4. hostile region descriptors (each must be refused or accepted, never crash)
base = generate_region(77, dims=(6, 5, 3), k_range=(3, 3), net_span=4)
rc, h = create(n_envs=2); assert rc == 0
def rec(t, used=0, net=-1, pin=-1): return int(pack_records([t], [used], [net], [pin])[0])
hostile = [dict(dims=(0, 5, 3)), dict(dims=(6, -1, 3)), dict(dims=(6, 5, 0)), dict(dims=(6, 5, 33)), dict(dims=(1 << 15, 1 << 15, 2)), dict(dims=(1 << 30, 4, 1)),
           dict(xs_host=None), dict(ys_host=None), dict(layer_dir_host=None), dict(nodes_host=None), dict(n_nets=-1), dict(n_nets=16383),
           dict(xs=base.xs[::-1].copy()), dict(ys=np.zeros_like(base.ys)),
           dict(xs=np.array([-2**31, -5, 0, 7, 2**31 - 2, 2**31 - 1], np.int32)), dict(ys=np.array([-2**31, 0, 1, 2, 2**31 - 1], np.int32)),
           dict(xs=(np.arange(6) * (1 << 26)).astype(np.int32)), dict(metrics0=(-2**31, 2**31 - 1, -1)),
           dict(nodes=np.full(90, rec(ACCESS, 0, 0, 0), np.uint32), n_nets=1),                         # 90 access points of one net (> 128? no: accepted)
           dict(dims=(6, 5, 5), xs=base.xs, ys=base.ys, layer_dir=np.ones(5, np.uint8), nodes=np.full(150, rec(ACCESS, 0, 0, 0), np.uint32), n_nets=1),   # 150 > 128
           dict(nodes=np.full(90, rec(ACCESS, 0, 5, 0), np.uint32), n_nets=3),                         # net id beyond n_nets
           dict(nodes=np.full(90, rec(ACCESS, 1, -1, -1), np.uint32)),                                 # ACCESS without a net
           dict(nodes=np.full(90, 0xFFFFFFFF, np.uint32), n_nets=16382), dict(nodes=np.full(90, 0xFFFFFFFF, np.uint32), n_nets=5),
           dict(nodes=np.zeros(90, np.uint32)), dict(layer_dir=np.full(3, 255, np.uint8))]
for kw in hostile:
    rc = load(h, [desc(base, **kw)]); assert rc <= 0, (kw, rc); note(rc)
    rc = load(h, [desc(base), desc(base, **kw)]); assert rc <= 0; note(rc)
for it in range(400):                                   # random records, random small dims, random coordinate tables
    X_, Y_, Z_ = (int(v) for v in rng.integers(1, 7, 3))
    n = X_ * Y_ * Z_
    kind = it % 4
    if kind == 0: nodes = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    elif kind == 1: nodes = pack_records(rng.integers(0, 3, n), rng.integers(0, 2, n), rng.integers(-1, 6, n), rng.integers(-1, 300, n))
    elif kind == 2: nodes = pack_records(np.full(n, ACCESS), rng.integers(0, 2, n), rng.integers(0, 3, n), rng.integers(0, 2, n))
    else: nodes = pack_records(rng.choice([BLOCKAGE, ACCESS], n), np.zeros(n, int), np.zeros(n, int), rng.integers(0, 9, n))
    xs = np.cumsum(rng.integers(1, [4, 1 << 20, 1 << 27][it % 3], X_)).astype(np.int32)
    ys = np.cumsum(rng.integers(1, 4000, Y_)).astype(np.int32)
    kw = dict(dims=(X_, Y_, Z_), xs=xs, ys=ys, layer_dir=rng.integers(0, 2, Z_).astype(np.uint8), nodes=nodes, n_nets=int(rng.choice([0, 1, 5, 7, 16382])))
    rc = load(h, [desc(base, **kw)]); assert rc <= 0; note(rc)
    if rc == 0:
        act = np.ones(2, np.int32); assert L.xr_batch_step(h, act.ctypes.data, None) == 0
# ---- 5. device out of memory at every allocation of a load: XR_ERR_NOMEM, nothing leaked, the batch reloads fine afterwards
regs = [desc(generate_region(5, dims=(7, 5, 3), k_range=(2, 4), net_span=4))]
for lim in (0, 8, 64, 200, 512, 1024, 2048, 4096, 1 << 14):
    limit.value = lim
    rc = load(h, regs); assert rc in (0, X.XR_ERR_NOMEM), (lim, rc); note(rc)
    assert L.xr_batch_step(h, p, None) in (0, X.XR_ERR_STATE)
limit.value = -1
assert load(h, regs) == 0
rc, h2 = create(n_envs=2, guide_cost=300, maze_end_iter=2); assert rc == 0 and load(h2, regs) == 0
o = np.array([0, 1, 1, 2, 2][: regs[0].n_nets + 1], np.int32); bx = np.zeros((4, 6), np.int16)
oa = (vp * 1)(o.ctypes.data); ba = (vp * 1)(bx.ctypes.data)
assert L.xr_batch_load_guides(h2, oa, ba, None) == 0
for lim in (0, 4, 16):
    limit.value = lim
    rc = L.xr_batch_load_guides(h2, oa, ba, None); assert rc in (0, X.XR_ERR_HIP, X.XR_ERR_NOMEM), rc; note(rc)
    assert L.xr_batch_step(h2, p, None) == 0
limit.value = -1
L.xr_batch_destroy(h); L.xr_batch_destroy(h2)
assert live.value == 0, ("leaked device bytes", live.value)
print("HOSTSAN_OK", n_loaded, sorted(counts.items()))
