#!/usr/bin/env python3
"""tools/ab_bench.py — same-box interleaved A/B of library builds (tools/build_variant.sh).

    python tools/ab_bench.py [--rounds 3] [--steps 30] [--check] NAME[=path] ...
`NAME` = a directory under gpurun_abl/ (or `tree` = the in-tree library). Every round runs bench.py's headline leg once per variant, in
order; prints cubes/s and the per-kernel ms of each run and the per-variant medians. --check: first compares each variant's outputs with
the first variant's on a small scene (tools/ab_outputs.py) and against nothing else — parity proper is tests/ -m gpu."""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lib_of(name):
    if "=" in name:
        name, path = name.split("=", 1)
        return name, path
    if name == "tree":
        return name, os.path.join(ROOT, "surfacenet_amd", "libsurfacenet_hip.so")
    return name, os.path.join(ROOT, "gpurun_abl", name, "libsurfacenet_hip.so")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--with-simil", action="store_true", help="also run bench.py's similarityNet leg (patches/s)")
    ap.add_argument("--extra", default="", help="extra bench.py arguments, e.g. '--precision f16'")
    ap.add_argument("--keys", default="merge_conv_b,merge_conv_a,conv1_2,conv2_2,conv3_2,conv4_2,side_op234_deconv")
    ap.add_argument("variants", nargs="+")
    args = ap.parse_args()
    variants = [lib_of(v) for v in args.variants]
    keys = args.keys.split(",")
    if args.check:
        outs = []
        for name, lib in list(variants):
            out = "/tmp/ab_%s.npz" % name
            p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ab_outputs.py"), "save", out], env=dict(os.environ, SURFACENET_HIP_LIB=lib),
                               capture_output=True, text=True)
            if p.returncode != 0:
                print("== %s FAILS the output check and is dropped:\n%s" % (name, p.stderr[-600:]), flush=True)
                variants.remove((name, lib))
                continue
            outs.append(out)
        for (name, _), out in zip(variants[1:], outs[1:]):
            print("== outputs of %s vs %s" % (name, variants[0][0]), flush=True)
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "ab_outputs.py"), "cmp", outs[0], out])
    res = {name: [] for name, _ in variants}
    for r in range(args.rounds):
        for name, lib in variants:
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(args.steps), "--warmup", "3", "--no-fast-mode", "--no-cpu-baseline", "--no-s64",
                   "--no-post-pass", "--no-scenes"] + ([] if args.with_simil else ["--no-simil"]) + args.extra.split()
            p = subprocess.run(cmd, env=dict(os.environ, SURFACENET_HIP_LIB=lib), capture_output=True, text=True)
            if p.returncode != 0:
                print("%s: bench.py failed:\n%s" % (name, p.stderr[-2000:]), flush=True)
                continue
            j = json.loads(p.stdout.strip().splitlines()[-1])
            k = j["kernels_ms_per_step"]
            res[name].append((j["value"], [k.get(x, float("nan")) for x in keys]))
            sim = ("  simil %.0f patches/s (s_conv1_2 %.3f ms)" % (j["similarity_net"]["value"], j["similarity_net"]["kernels_ms_per_step"].get("s_conv1_2", float("nan")))) if "similarity_net" in j else ""
            print("round %d %-12s %8.1f cubes/s  %s%s" % (r, name, j["value"], "  ".join("%s %.3f" % (x, k.get(x, float("nan"))) for x in keys), sim), flush=True)
    print("== medians")
    for name, _ in variants:
        if not res[name]:
            continue
        med = statistics.median(v for v, _ in res[name])
        km = [statistics.median(ks[i] for _, ks in res[name]) for i in range(len(keys))]
        print("%-12s %8.1f cubes/s  %s" % (name, med, "  ".join("%s %.3f" % (x, v) for x, v in zip(keys, km))), flush=True)


if __name__ == "__main__":
    main()
