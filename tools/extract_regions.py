#!/usr/bin/env python3
"""Extract per-GCell regions from a LEF/DEF/guide triple (xroute_env_amd/lefdef.py) into a compact region pack.

    python tools/extract_regions.py --out tests/golden/ispd18_test1_regions.npz --count 256 --stride 12     # (the committed pack: --stride 16 yields 206 regions)

Default inputs are the reference's ispd18_test1 files (build container only).  The pack is DATA derived from the
reference's benchmark input files (tracks, placed pin / obstruction shapes, guides), not source."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xroute_env_amd import lefdef

REF = "/root/reference/ispd/ispd18_test1/ispd18_test1.input"
ap = argparse.ArgumentParser()
ap.add_argument("--lef", default=REF + ".lef")
ap.add_argument("--def", dest="deff", default=REF + ".def")
ap.add_argument("--guide", default=REF + ".guide")
ap.add_argument("--out", required=True)
ap.add_argument("--count", type=int, default=256)
ap.add_argument("--stride", type=int, default=16, help="keep every stride-th non-empty region (spreads over the die)")
ap.add_argument("--min-nets", type=int, default=2)
ap.add_argument("--static-region1", default=None, metavar="OUT",
                help="also write the one region the reference describes (xroute_env/__init__.py:13-23: ISPD-2018 test1, 1x1, position "
                     "(199500, 245100)-(205200, 250800)) as a one-region pack: xroute_env_amd/data/static_regions.npz")
args = ap.parse_args()
t0 = time.time()
design = lefdef.load_design(args.lef, args.deff, args.guide)
ex = lefdef.RegionExtractor(design)
if args.static_region1:
    rb = (199500, 245100, 205200, 250800)
    lefdef.save_region_pack(args.static_region1, [ex.extract((rb[0] - 2000, rb[1] - 2000, rb[2] + 2000, rb[3] + 2000), name="region1", route_box=rb)])
regs = ex.gcell_regions(min_nets=args.min_nets)
keep = regs[::args.stride][:args.count]
lefdef.save_region_pack(args.out, keep)
print(f"{len(regs)} regions with >= {args.min_nets} nets on the die, kept {len(keep)} -> {args.out} "
      f"({os.path.getsize(args.out) / 1024:.0f} KiB) in {time.time() - t0:.0f}s; "
      f"mean K {sum(r.n_nets for r in keep) / len(keep):.1f}, dims of the first: {keep[0].dims}")
