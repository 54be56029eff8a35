#!/usr/bin/env python3
"""oracle/gen_golden_w5d.py — TEST INFRASTRUCTURE. Pins the one constant of the 3D-CNN that CAN be pinned without Theano: the fixed
"bilinear" interpolation kernel `__W_5D__` of nets/layers.py:361-372, by EXECUTING the reference's own function.

nets/layers.py cannot be imported here (it imports theano / lasagne at module level), but `__W_5D__` is pure numpy: the function
is located in the file's AST, compiled on its own and called with the two kernel sizes the network uses
(k_size = upscale_factor/2*2+1 under Python-2 integer division, nets/layers.py:383: f=2 -> 3, f=4 -> 5). Only the resulting
arrays are written (tests/golden/w5d_cases.npz); no source text.

Usage:  python oracle/gen_golden_w5d.py   (from the repo root; needs /root/reference)
"""
import ast
import os

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def load_w5d():
    path = os.path.join(REF, "nets", "layers.py")
    tree = ast.parse(open(path).read(), filename=path)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "__W_5D__"]
    assert len(fn) == 1
    mod = ast.Module(body=fn, type_ignores=[])
    ns = {"np": np}
    exec(compile(mod, path, "exec"), ns)
    return ns["__W_5D__"]


def main():
    w5d = load_w5d()
    out = {}
    for f in (2, 4):
        k = f // 2 * 2 + 1                       # nets/layers.py:383 under Python-2 integer division
        W = w5d(k)
        assert W.shape == (1, 1, k, k, k) and W.dtype == np.float32
        out["f%d_k" % f] = np.int64(k)
        out["f%d_W" % f] = W
    np.savez(os.path.join(OUT, "w5d_cases.npz"), **out)
    print({k: (v.shape if getattr(v, "ndim", 0) else int(v)) for k, v in out.items()}, out["f2_W"][0, 0, 1], out["f4_W"][0, 0, 2, 2])


if __name__ == "__main__":
    main()
