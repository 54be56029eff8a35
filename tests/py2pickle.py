"""Writes the byte stream Python 2.7 + numpy 1.13 produce for `cPickle.dump(list_of_ndarrays, f, protocol=2)` — the format of the
reference's `*.model` files (nets/SurfaceNet.py:397-400, nets/similarityNet.py:240-242; conda_list_explicit.txt: python 2.7.13,
numpy 1.13.1) — without a Python 2 interpreter: the opcodes are emitted by hand.

What makes it a *Python-2* pickle (and what `weights.load_lasagne_pickle` must therefore cope with under Python 3):
  * the array constructor is the global `numpy.core.multiarray._reconstruct` (numpy >= 2 moved the module);
  * dtype strings and the raw array bytes are py2 `str` objects: SHORT_BINSTRING / BINSTRING opcodes, which Python 3 can only
    read with `encoding='latin1'` (numpy then re-encodes the data string to bytes);
  * the dtype state tuple is the version-3 form `(3, '<', None, None, None, -1, -1, 0)`.
"""
import struct

import numpy as np

PROTO, GLOBAL, MARK, TUPLE, REDUCE, BUILD, BININT1, BININT, NONE, NEWFALSE, EMPTY_LIST, APPENDS, STOP = (
    b"\x80\x02", b"c", b"(", b"t", b"R", b"b", b"K", b"J", b"N", b"\x89", b"]", b"e", b".")
SHORT_BINSTRING, BINSTRING, TUPLE1, TUPLE2, TUPLE3, BINPUT, LONG_BINPUT = b"U", b"T", b"\x85", b"\x86", b"\x87", b"q", b"r"


def _int(v):
    return BININT1 + bytes([v]) if 0 <= v < 256 else BININT + struct.pack("<i", v)


def _str(b):
    return SHORT_BINSTRING + bytes([len(b)]) + b if len(b) < 256 else BINSTRING + struct.pack("<i", len(b)) + b


def _array(a, memo):
    a = np.ascontiguousarray(a)
    descr = a.dtype.str.lstrip("<>|=").encode()                # 'f4'
    out = GLOBAL + b"numpy.core.multiarray\n_reconstruct\n" + memo()
    out += GLOBAL + b"numpy\nndarray\n" + memo()
    out += _int(0) + TUPLE1 + memo() + _str(b"b") + memo() + TUPLE3 + memo() + REDUCE + memo()     # _reconstruct(ndarray, (0,), 'b')
    out += MARK + _int(1)                                                                          # state = (1, shape, dtype, fortran, data)
    out += (MARK + b"".join(_int(int(d)) for d in a.shape) + TUPLE if a.ndim > 3 else
            b"".join(_int(int(d)) for d in a.shape) + (b")", TUPLE1, TUPLE2, TUPLE3)[a.ndim]) + memo()
    out += GLOBAL + b"numpy\ndtype\n" + memo() + _str(descr) + memo() + _int(0) + _int(1) + TUPLE3 + memo() + REDUCE + memo()
    out += MARK + _int(3) + _str(b"<" if a.dtype.itemsize > 1 else b"|") + memo() + NONE + NONE + NONE + BININT + struct.pack("<i", -1) + \
        BININT + struct.pack("<i", -1) + _int(0) + TUPLE + memo() + BUILD
    out += NEWFALSE + _str(a.tobytes()) + memo() + TUPLE + memo() + BUILD
    return out


def dumps_py2(arrays):
    """bytes of cPickle.dumps(list(arrays), protocol=2) under Python 2.7 / numpy 1.13 (memo indices as cPickle assigns them,
    one per constructed object; repeated globals are re-emitted instead of memo-fetched, which every unpickler accepts)."""
    n = [0]

    def memo():
        i = n[0]
        n[0] += 1
        return BINPUT + bytes([i]) if i < 256 else LONG_BINPUT + struct.pack("<i", i)

    out = PROTO + EMPTY_LIST + memo() + MARK
    for a in arrays:
        out += _array(a, memo)
    return out + APPENDS + STOP
