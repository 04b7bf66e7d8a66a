"""Drive tools/sim/dial_sim.c against the oracle on BASELINE config regions: exactness + round statistics."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import xr_oracle as orc
from xroute_env_amd.regions import config_regions, unpack_records, ACCESS, BLOCKAGE

class Stats(C.Structure):
    _fields_ = [(n, C.c_long) for n in ("rounds", "expansions", "relax_ok", "searches", "max_open", "scans", "routes")]

L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdial_sim.so"))
vp = C.c_void_p
L.dial_route.argtypes = [C.c_int]*3 + [vp]*5 + [C.c_int, vp, vp] + [C.c_int]*4 + [vp, vp, C.c_int, vp, vp]
C.c_int.in_dll(L, 'g_astar').value = int(os.environ.get('ASTAR', '0'))
C.c_int.in_dll(L, 'g_chain').value = int(os.environ.get('CHAIN', '0'))
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
nreg = int(sys.argv[2]) if len(sys.argv) > 2 else 16
mults = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 4]
regions = config_regions(cfg, nreg)
for mult in mults:
    st = Stats()
    bad = 0
    for r in regions:
        env = orc.OracleEnv(r)
        X, Y, Z = r.dims
        ntype, used, net, pin = unpack_records(r.nodes)
        node_net = np.where(ntype == ACCESS, net + 1, np.where(ntype == BLOCKAGE, -1, 0)).astype(np.int16)
        rng = np.random.default_rng(1)
        maxr = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 30
        nr = 0
        while env.nlegal() > 0 and nr < maxr:
            nr += 1
            legal = env.legal()
            a = int(legal[rng.integers(0, len(legal))])
            owner = env.owner().copy()
            aps = np.flatnonzero(node_net == a).astype(np.int32)
            appin = (pin[aps] + 1).astype(np.int32)
            ref = env.step(a)
            delta = np.zeros(3, np.int32); path = np.zeros(r.n_nodes, np.int32); plen = C.c_int(0)
            xs = np.ascontiguousarray(r.xs, np.int32); ys = np.ascontiguousarray(r.ys, np.int32); ld = np.ascontiguousarray(r.layer_dir, np.uint8)
            rc = L.dial_route(X, Y, Z, xs.ctypes.data, ys.ctypes.data, ld.ctypes.data, node_net.ctypes.data, owner.ctypes.data, a,
                              aps.ctypes.data, appin.ctypes.data, len(aps), 800, 3200, mult, delta.ctypes.data, path.ctypes.data,
                              r.n_nodes, C.byref(plen), C.byref(st))
            ok = rc == 0 and delta.tolist() == ref["delta"].tolist() and path[:plen.value].tolist() == ref["path"].tolist() \
                and np.array_equal(owner, env.owner())
            bad += not ok
    print(f"config {cfg} mult {mult}: routes {st.routes} searches {st.searches} mismatches {bad}; per search: rounds {st.rounds/max(st.searches,1):.1f} "
          f"expansions {st.expansions/max(st.searches,1):.0f} relax {st.relax_ok/max(st.searches,1):.0f}; per route: rounds {st.rounds/st.routes:.1f} "
          f"expansions {st.expansions/st.routes:.0f}; max open {st.max_open}")
