#!/usr/bin/env python3
"""tools/linf_survey.py — device L_inf of the surface probabilities against the fp64 oracle over MANY inputs (GPU box; TEST INFRASTRUCTURE: uses
oracle/ and tests/ as the checker). The parity tests assert a tolerance on a handful of seeded cases; this prints the distribution behind it, per input
family (tests/survey_inputs.py: noise / structured / real-pixel windows shifted and flipped / scene cubes of both grids incl. border cubes, N_vp = 2, 5,
16 / the x3.2 stress net calibrated), all three BN-calibrated test nets:
    python tools/linf_survey.py [--out gpurun_out/linf_survey.json] [--limit N] [--families noise,scene]
Every case is replayable by (family, key) - the worst ones are pinned in tests/test_gpu_parity.py::test_survey_worst_cases.
With SURFACENET_HIP_LIB=<test-only twin> and SN_C4_M8=0 the same inputs run the round-4 arithmetic (conv4 chain on three fp16 MFMAs)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run_case(surfacenet_amd, net_ctx, cvc_ctx_for, family, key):
    """-> (label, L_inf, n_samples): the device's unfused probabilities against the fp64 oracle on the case's own input."""
    import survey_inputs
    from oracle import net_oracle
    net, stress, X, label = survey_inputs.make_case(family, key, cvc_ctx_for)
    ctx, values = net_ctx(net, stress)
    unf = []
    if stress:
        ctx.load_param_values(values)                                  # static exponents again
        ctx.forward(X, None, n_vp=1)
        ctx.calibrate(X.shape[0], max_sat_fraction=1e-3)
        ctx.numeric_status()
    for i in range(0, X.shape[0], ctx.max_samples):
        unf.append(ctx.forward(X[i:i + ctx.max_samples], None, n_vp=1)[1])
    unf = np.concatenate(unf)
    _, u64 = net_oracle.forward_torch(X, values, n_vp=1)
    return label, float(np.abs(unf - u64).max()), int(X.shape[0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--limit", type=int, default=0, help="only the first N cases of every family")
    ap.add_argument("--families", default="")
    ap.add_argument("--replay", default="", help="a previous --out file: run its --top worst non-stress cases again (e.g. under SURFACENET_HIP_LIB=<twin> SN_C4_M8=0)")
    ap.add_argument("--top", type=int, default=8)
    a = ap.parse_args()
    import surfacenet_amd
    import survey_inputs
    nets, cvcs = {}, {}

    def net_ctx(net, stress):
        k = (net, stress)
        if k not in nets:
            c = surfacenet_amd.Context(cube_D=survey_inputs.S, max_samples=16)
            v = survey_inputs.net_values(net, stress)
            c.load_param_values(v)
            nets[k] = (c, v)
        return nets[k]

    def cvc_ctx_for(tag, P, imgs):
        if tag not in cvcs:
            c = surfacenet_amd.Context(cube_D=survey_inputs.S, max_samples=16)
            c.set_cameras(P); c.set_images(imgs)
            cvcs[tag] = c
        return cvcs[tag]

    cases = survey_inputs.case_list()
    if a.replay:
        prev = [r for r in json.load(open(a.replay))["cases"] if r["family"] != "stress"]
        prev.sort(key=lambda r: -r["linf"])
        as_key = lambda k: tuple(k)
        cases = [(r["family"], as_key(r["key"])) for r in prev[:a.top]]
        print("replaying the %d worst non-stress cases of %s (there: %s)" % (len(cases), a.replay, ", ".join("%.3e" % r["linf"] for r in prev[:a.top])))
    fams = [f for f in a.families.split(",") if f]
    seen, rows, t0 = {}, [], time.time()
    for family, key in cases:
        if fams and family not in fams:
            continue
        seen[family] = seen.get(family, 0) + 1
        if a.limit and seen[family] > a.limit:
            continue
        label, e, ns = run_case(surfacenet_amd, net_ctx, cvc_ctx_for, family, key)
        rows.append({"family": family, "key": list(key), "label": label, "linf": e, "samples": ns})
        print("%-72s L_inf %.3e" % (label, e), flush=True)
    print("== %d cases, %d network inputs, %.0f s" % (len(rows), sum(r["samples"] for r in rows), time.time() - t0))
    summary = {}
    groups = {}
    for r in rows:
        groups.setdefault("scene*" if r["family"].startswith("scene") else r["family"], []).append(r["linf"])
    groups["ALL"] = [r["linf"] for r in rows]
    for g, v in groups.items():
        v = np.asarray(v)
        summary[g] = {"n": int(v.size), "median": float(np.median(v)), "p99": float(np.percentile(v, 99)), "max": float(v.max())}
        print("== %-12s n %3d  median %.3e  99 %% %.3e  max %.3e" % (g, v.size, np.median(v), np.percentile(v, 99), v.max()))
    worst = sorted(rows, key=lambda r: -r["linf"])[:5]
    for r in worst:
        print("   worst: %-66s %.3e   (%s, %r)" % (r["label"], r["linf"], r["family"], tuple(r["key"])))
    if a.out:
        json.dump({"cases": rows, "summary": summary, "lib": os.environ.get("SURFACENET_HIP_LIB", "product"), "SN_C4_M8": os.environ.get("SN_C4_M8", "")},
                  open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
