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


# ---- protocol 0 (what `pickle.dump(values, f)` writes under Python 2 when no protocol is given; the reference re-opens the file in
# TEXT mode, nets/SurfaceNet.py:398, which only an ASCII pickle survives on every platform) ---------------------------------------
def _repr_py2_str(b):
    """Python 2's repr() of a byte string: quote choice and escapes as stringobject.c:PyString_Repr."""
    quote = b'"' if (b"'" in b and b'"' not in b) else b"'"
    out = bytearray(quote)
    q = quote[0]
    for c in b:
        if c == q or c == 0x5C:
            out += b"\\" + bytes([c])
        elif c == 0x09:
            out += b"\\t"
        elif c == 0x0A:
            out += b"\\n"
        elif c == 0x0D:
            out += b"\\r"
        elif c < 0x20 or c >= 0x7F:
            out += b"\\x%02x" % c
        else:
            out.append(c)
    return bytes(out + quote)


def dumps_py2_proto0(arrays):
    """bytes of pickle.dumps(list(arrays)) (protocol 0) under Python 2.7 / numpy 1.13: GLOBAL / MARK / INT / STRING / TUPLE / REDUCE /
    BUILD / PUT / APPEND opcodes in their text forms, the array bytes as a repr()-escaped py2 `str`."""
    n = [0]

    def put():
        i = n[0]
        n[0] += 1
        return b"p%d\n" % i

    I = lambda v: b"I%d\n" % int(v)
    S = lambda b: b"S" + _repr_py2_str(b) + b"\n"
    out = b"(l" + put()
    for a in arrays:
        a = np.ascontiguousarray(a)
        descr = a.dtype.str.lstrip("<>|=").encode()
        out += b"cnumpy.core.multiarray\n_reconstruct\n" + put() + b"(cnumpy\nndarray\n" + put() + b"(" + I(0) + b"t" + put() + S(b"b") + put() + b"t" + put() + b"R" + put()
        out += b"(" + I(1) + b"(" + b"".join(I(d) for d in a.shape) + b"t" + put()
        out += b"cnumpy\ndtype\n" + put() + b"(" + S(descr) + put() + I(0) + I(1) + b"t" + put() + b"R" + put()
        out += b"(" + I(3) + S(b"<" if a.dtype.itemsize > 1 else b"|") + put() + b"NNN" + I(-1) + I(-1) + I(0) + b"t" + put() + b"b"
        out += b"I00\n" + S(a.tobytes()) + put() + b"t" + put() + b"b" + b"a"
    return out + b"."
