"""One-off: hand-parse the reference's only recorded simulator output — the TensorBoard event file of its PPO run
(baseline/PPO/results/2023-04-27--05-00-38/events.out.tfevents.*) — into tests/golden/g8_ppo_episode_stats.json.

Run in the BUILD container only (needs /root/reference).  No tensorflow: a TFRecord is  u64 length | u32 crc | payload | u32 crc ;
the payload is an `Event` protobuf (wall_time = 1: double, step = 2: varint, summary = 5: Summary{ repeated value = 1:
{ tag = 1: string, simple_value = 2: float } }).  The fixture holds per-tag summary statistics (data, not source)."""
import json
import os
import struct
import sys

REF = "/root/reference/baseline/PPO/results/2023-04-27--05-00-38"


def varint(buf, i):
    v = s = 0
    while True:
        b = buf[i]; i += 1
        v |= (b & 0x7F) << s
        s += 7
        if not b & 0x80:
            return v, i


def fields(buf):
    i = 0
    while i < len(buf):
        key, i = varint(buf, i)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, i = varint(buf, i)
        elif wt == 1:
            v = buf[i:i + 8]; i += 8
        elif wt == 2:
            n, i = varint(buf, i)
            v = buf[i:i + n]; i += n
        elif wt == 5:
            v = buf[i:i + 4]; i += 4
        else:
            raise ValueError(f"wire type {wt}")
        yield fn, wt, v


def main():
    path = [os.path.join(REF, f) for f in os.listdir(REF) if f.startswith("events.out.tfevents")][0]
    data = open(path, "rb").read()
    i = 0
    series = {}
    t0 = t1 = None
    while i + 12 <= len(data):
        (n,) = struct.unpack("<Q", data[i:i + 8])
        payload = data[i + 12:i + 12 + n]
        i += 12 + n + 4
        wall = step = None
        vals = []
        for fn, wt, v in fields(payload):
            if fn == 1 and wt == 1:
                (wall,) = struct.unpack("<d", v)
            elif fn == 2 and wt == 0:
                step = v
            elif fn == 5 and wt == 2:
                for f2, w2, v2 in fields(v):
                    if f2 == 1 and w2 == 2:
                        tag = val = None
                        for f3, w3, v3 in fields(v2):
                            if f3 == 1 and w3 == 2:
                                tag = v3.decode()
                            elif f3 == 2 and w3 == 5:
                                (val,) = struct.unpack("<f", v3)
                        if tag is not None and val is not None:
                            vals.append((tag, val))
        if wall is not None:
            t0 = wall if t0 is None else min(t0, wall)
            t1 = wall if t1 is None else max(t1, wall)
        for tag, val in vals:
            series.setdefault(tag, []).append((step, val))
    out = {"source": "baseline/PPO/results/2023-04-27--05-00-38/events.out.tfevents.* (hand-parsed TFRecord, tools/parse_ppo_tfevents.py)",
           "wall_seconds": (t1 - t0) if t0 is not None else None, "tags": {}}
    for tag, pts in sorted(series.items()):
        v = [p[1] for p in pts]
        out["tags"][tag] = {"count": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v),
                            "first_step": pts[0][0], "last_step": pts[-1][0]}
    # env steps of the run: PPO updates every 100 env steps (baseline/PPO/train_PPO.py:18 update_timestep) with K_epochs = 10 loss
    # points per update (baseline/PPO/train_PPO.py:19), so loss points / 10 * 100 = env steps
    nloss = out["tags"].get("2.Training/1.Loss", {}).get("count", 0)
    nep = out["tags"]["1.Episode/1.reward"]["count"]
    out["derived"] = {"env_steps": nloss // 10 * 100, "episodes": nep, "steps_per_episode": (nloss // 10 * 100) / nep,
                      "note": "steps per episode = nets routed per episode (one Game.step per net, baseline/PPO/train_PPO.py:96-99)"}
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g8_ppo_episode_stats.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1)[:1500])


if __name__ == "__main__":
    sys.exit(main())
