"""Obstacle tower alone: time per launch and agreement of library builds on the same input.
python tools/tower_probe.py [envs] [D H W] — libraries from XR_TOWER_LIBS (comma separated, default libxroute_hip.so); one subprocess per
library (the loader binds one build per process), outputs compared in the parent."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(B, dims, out_path):
    import torch
    from xroute_env_amd import agents
    torch.manual_seed(3)
    dev = torch.device("cuda:0")
    rep = agents.RepresentationNetwork().to(dev).eval()
    with torch.no_grad():                       # BatchNorm statistics away from the identity, as a trained network has them
        for m in rep.modules():
            if isinstance(m, torch.nn.BatchNorm3d):
                m.running_mean.normal_(0, 0.3); m.running_var.uniform_(0.5, 2.0); m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2)
    D, H, W = dims
    tower = agents.FusedObstacleTower(rep, dims, dev)
    assert tower.supported, dims
    g = torch.Generator(device="cpu").manual_seed(5)
    head = (torch.rand((B, D * H * W), generator=g) < 0.3).float().to(dev)
    out = tower(head)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(3): tower(head)
    ev[0].record()
    for _ in range(20): out = tower(head)
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 20
    with torch.no_grad():
        ref = agents.normalize(rep.encode_obstacles(head[:64].reshape(64, 1, D, H, W)))
    err = float((out[:64] - ref).abs().max())
    torch.save(out.cpu(), out_path)
    if os.environ.get("XT_PHASES"):             # a -DXT_PHASE_TIMING build (make ttiming): thread 0's cycles per stage in the first floats of a row
        names = ["zero + load x", "block(1) conv1", "block(1) conv2 + residual", "align1 5x5x5 1->7", "block(7) conv1 (MFMA)", "block(7) conv2 + align2 sums (MFMA)", "final sums + normalise"]
        cyc = out[:, :7].double().mean(0).tolist()
        tot = sum(cyc)
        for n_, c in zip(names, cyc):
            print(f"  {n_:40s} {c:9.0f} cycles {100 * c / tot:5.1f}%")
        print(f"  {'total':40s} {tot:9.0f} cycles per env (thread 0)")
        print("  block(7) conv1, cycles of every wave's fill loop + tiles:", " ".join(f"{v / 1000:.1f}k" for v in out[:, 16:32].double().mean(0).tolist()))
        print("  align1, cycles of every wave's tiles:", " ".join(f"{v / 1000:.1f}k" for v in out[:, 32:48].double().mean(0).tolist()))
        print("  align1, then the next stage's operands arrive:", " ".join(f"{v / 1000:.1f}k" for v in out[:, 48:64].double().mean(0).tolist()))
        sub = out[:, 7:11].double().mean(0).tolist()
        print(f"  block(7) conv1, wave 0: fill loop {sub[0]:.0f}, its own tiles {sub[1]:.0f}, the next convolution's operands {sub[2]:.0f}, waiting at the barrier {sub[3]:.0f} cycles")
    print(json.dumps({"lib": os.environ.get("XR_LIB", "libxroute_hip.so"), "envs": B, "dims": dims, "ms_per_launch": round(ms, 4),
                      "ms_per_1024_envs": round(ms * 1024 / B, 4), "max_abs_err_vs_framework_path": err}))


if __name__ == "__main__":
    if sys.argv[1:2] == ["--child"]:
        child(int(sys.argv[2]), tuple(int(v) for v in sys.argv[3:6]), sys.argv[6])
        sys.exit(0)
    a = [int(v) for v in sys.argv[1:]]
    B = a[0] if a else 1024
    dims = tuple(a[1:4]) if len(a) >= 4 else (9, 40, 24)
    libs = os.environ.get("XR_TOWER_LIBS", "libxroute_hip.so").split(",")
    outs = []
    for rep_ in range(2):
        for i, lib in enumerate(libs):
            path = f"/tmp/tower_probe_{i}.pt"
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(B)] + [str(v) for v in dims] + [path],
                               env=dict(os.environ, XR_LIB=lib), capture_output=True, text=True)
            print(r.stdout.strip() or r.stderr[-2000:])
    if len(libs) > 1:
        import torch
        o = [torch.load(f"/tmp/tower_probe_{i}.pt") for i in range(len(libs))]
        for i in range(1, len(libs)):
            print(f"{libs[i]} vs {libs[0]}: max abs diff {float((o[i] - o[0]).abs().max()):.3e}")
