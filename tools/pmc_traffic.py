#!/usr/bin/env python3
"""tools/pmc_traffic.py — turns a tools/pmc_summary.py summary into profiles/pmc_traffic.json, the file bench.py reads `roofline.traffic` from.

    python tools/pmc_traffic.py --summary profiles/r2/summary_r2x.json --cube-d 32 --samples 128 --precision f16x3

The counters behind the summary must come from rocprofv3 --pmc passes of the HEADLINE workload only
(`bench.py --no-fast-mode --no-cpu-baseline --no-s64 --no-simil --no-post-pass`): the other legs launch the same kernels with other
arguments (e.g. the post-pass leg makes cvc_warp_kernel write the planar fp32 tensor as well, 56 instead of 32 bytes per voxel) and
would be averaged in. The file is stamped with the hash of the kernel sources it was measured on (bench.kernel_src_sha16); bench.py
drops the traffic figure when the sources have changed since."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--summary", required=True)
    ap.add_argument("--cube-d", type=int, default=32)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--precision", default="f16x3")
    ap.add_argument("--layers", nargs="*", default=["merge_conv_b=3,1,8,7,1,2,1,2,4,0,0,-1", "merge_conv_a=3,1,8,7,0,2,1,2,4,0,0,-1"],      # (round 4: the 4-wave PWM instantiations)
                    help="layer=template-argument list of its conv3d_f16_mfma instantiation")
    a = ap.parse_args()
    import bench
    summ = json.load(open(a.summary))["kernels"]
    out = {"_note": "HBM-side bytes per launch from rocprofv3 PMC passes of the headline workload (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, one "
                    "counter set per pass; Infinity-Cache hits are included in FETCH_SIZE); source %s" % os.path.relpath(a.summary, ROOT),
           "kernel_src_sha16": bench.kernel_src_sha16(), "config": {"cube_D": a.cube_d, "samples": a.samples, "precision": a.precision}}
    for spec in a.layers:
        layer, targs = spec.split("=")
        k = "conv3d_f16_mfma<%s>" % targs
        if k in summ and "hbm_read_bytes_per_launch" in summ[k]:
            out[layer] = {"read_bytes": summ[k]["hbm_read_bytes_per_launch"], "write_bytes": summ[k].get("hbm_write_bytes_per_launch", 0.0),
                          "l2_hit_rate": summ[k].get("l2_hit_rate"), "mfma_util": summ[k].get("mfma_util"), "kernel": k}
    if "cvc_warp_kernel" in summ and "hbm_read_bytes_per_launch" in summ["cvc_warp_kernel"]:
        out["cvc_warp"] = {"read_bytes": summ["cvc_warp_kernel"]["hbm_read_bytes_per_launch"], "write_bytes": summ["cvc_warp_kernel"].get("hbm_write_bytes_per_launch", 0.0),
                           "kernel": "cvc_warp_kernel"}
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print("wrote profiles/pmc_traffic.json for kernel sources", out["kernel_src_sha16"], "layers", [k for k in out if k not in ("_note", "kernel_src_sha16", "config")])


if __name__ == "__main__":
    main()
