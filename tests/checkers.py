"""ctypes bindings for the two CHECKERS (test infrastructure only):

* ``Oracle``  -- oracle/libcc_oracle.so, the repo's plain-C restatement.
* ``RefLib``  -- oracle/_ref/libccref_o{0,1}.so, the real reference compiled by
  oracle/Makefile (present only where it has been built; it travels to the GPU
  box as a prebuilt file, /root/reference itself does not).

Nothing in the product package imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
P = C.c_void_p

BCH, RS = 0, 1
PGZ, BM, EUKLID = 0, 1, 2
MS, NMS, OMS, SCMS1, SCMS2, NMS2D = 0, 1, 2, 3, 4, 5
O0, O1, O2 = 0, 1, 2
ALG_NAMES = {PGZ: "PGZ", BM: "BM", EUKLID: "EUKLID"}
SOFT_NAMES = {MS: "MS", NMS: "NMS", OMS: "OMS", SCMS1: "SCMS1", SCMS2: "SCMS2", NMS2D: "2DNMS"}

# Variant ids of oracle/ref_driver.cc -> (oracle variant, alpha, beta) as the
# reference's tag types define them (soft_decision.h:36-73; Q11: the 2D tag's
# beta is Beta::num / Alpha::den).
REF_VARIANTS = {
    0: (MS, 1.0, 0.0),
    1: (NMS, 8 / 10, 0.0),
    2: (OMS, 1.0, 1 / 100),
    3: (SCMS1, 1.0, 0.0),
    4: (SCMS2, 1.0, 0.0),
    5: (NMS2D, 1.0, 1.0),
    6: (NMS2D, 3 / 4, 9 / 4),
    7: (NMS, 3 / 4, 0.0),
    8: (OMS, 1.0, 15 / 100),
}
# ids of the codes instantiated in oracle/ref_driver.cc: (family, q, t)
REF_CODES = {
    0: (BCH, 4, 2), 1: (BCH, 4, 3), 2: (BCH, 4, 2), 3: (BCH, 4, 2), 4: (BCH, 5, 3),
    5: (BCH, 6, 3), 6: (BCH, 8, 3), 7: (RS, 3, 1), 8: (RS, 3, 2), 9: (RS, 4, 3),
    10: (RS, 8, 16), 11: (BCH, 7, 2), 12: (BCH, 5, 2), 13: (BCH, 6, 4), 14: (RS, 8, 40), 15: (BCH, 8, 40),
}


def _ptr(a):
    return None if a is None else a.ctypes.data_as(P)


def build_oracle():
    """(Re)build oracle/libcc_oracle.so -- building the checker is not using it."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "oracle"], check=True)


class OrcCode(C.Structure):
    _fields_ = [
        ("family", C.c_int), ("q", C.c_int), ("t", C.c_int), ("n", C.c_int), ("k", C.c_int), ("l", C.c_int),
        ("dmin", C.c_int), ("mu", C.c_int), ("step", C.c_int), ("coding", C.c_int), ("size", C.c_int),
        ("exp_", C.c_uint8 * 512), ("log_", C.c_uint8 * 512),
        ("g", C.c_uint8 * 256), ("glen", C.c_int), ("h", C.c_uint8 * 256), ("hlen", C.c_int),
        ("roots", C.c_uint8 * 256), ("nroots", C.c_int),
    ]


class Oracle:
    """One code of the plain-C oracle."""
    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            path = os.path.join(ORACLE_DIR, "libcc_oracle.so")
            if not os.path.exists(path):
                build_oracle()
            cls._lib = C.CDLL(path)
            assert cls._lib.orc_code_sizeof() == C.sizeof(OrcCode)
        return cls._lib

    def __init__(self, family, q, t, mu=1, step=1, coding=0):
        self.c = OrcCode()
        rc = self.lib().orc_code_init(C.byref(self.c), family, q, t, mu, step, coding)
        if rc != 0:
            raise ValueError("orc_code_init failed: %d" % rc)
        for name in ("family", "q", "t", "n", "k", "l", "dmin"):
            setattr(self, name, getattr(self.c, name))
        self.g = np.array(self.c.g[: self.c.glen], dtype=np.uint8)
        self.h = np.array(self.c.h[: self.c.hlen], dtype=np.uint8)
        self.roots = np.array(self.c.roots[: self.c.nroots], dtype=np.uint8)
        self.exp = np.array(self.c.exp_[:], dtype=np.uint8)
        self.log = np.array(self.c.log_[:], dtype=np.uint8)

    def H(self):
        H = np.zeros((self.k, self.n), np.uint8)
        self.lib().orc_get_H(C.byref(self.c), _ptr(H))
        return H

    def H_alt(self):
        H = np.zeros((self.t * self.q, self.n), np.uint8)
        rows = C.c_int()
        self.lib().orc_get_H_alt(C.byref(self.c), _ptr(H), C.byref(rows))
        return H[: rows.value]

    @classmethod
    def minsum_H(cls, H, variant, iterations, y, alpha=1.0, beta=0.0, stop=O2):
        """min_sum__ over an explicit rows x cols matrix (no code involved)."""
        H = np.ascontiguousarray(H, np.uint8)
        rows, cols = H.shape
        y = np.ascontiguousarray(y, np.float32).reshape(-1, cols)
        B = y.shape[0]
        b = np.zeros((B, cols), np.uint8)
        L = np.zeros((B, cols), np.float32)
        iters = np.zeros(B, np.uint32)
        status = np.zeros(B, np.int32)
        it = C.c_uint()
        for i in range(B):
            status[i] = cls.lib().orc_minsum_H(_ptr(H), rows, cols, variant, iterations, C.c_double(alpha),
                                               C.c_double(beta), stop, _ptr(y[i]), _ptr(b[i]), _ptr(L[i]),
                                               C.byref(it))
            iters[i] = it.value
        return b, L, iters, status

    def to_string(self, alg_name):
        buf = C.create_string_buffer(128)
        self.lib().orc_to_string(C.byref(self.c), alg_name.encode(), buf, 128)
        return buf.value.decode()

    def encode(self, msg):
        msg = np.ascontiguousarray(msg, np.uint8)
        single = msg.ndim == 1
        msg = msg.reshape(-1, self.l)
        cw = np.zeros((msg.shape[0], self.n), np.uint8)
        for i in range(msg.shape[0]):
            rc = self.lib().orc_encode(C.byref(self.c), _ptr(msg[i]), _ptr(cw[i]))
            assert rc == 0, rc
        return cw[0] if single else cw

    def extract(self, cw):
        cw = np.ascontiguousarray(cw, np.uint8).reshape(-1, self.n)
        out = np.zeros((cw.shape[0], self.l), np.uint8)
        for i in range(cw.shape[0]):
            self.lib().orc_extract(C.byref(self.c), _ptr(cw[i]), _ptr(out[i]))
        return out

    def syndromes(self, b):
        b = np.ascontiguousarray(b, np.uint8)
        S = np.zeros(self.c.nroots, np.uint8)
        self.lib().orc_syndromes(C.byref(self.c), _ptr(b), _ptr(S))
        return S

    def locator(self, alg, S, erasures=()):
        er = np.asarray(erasures, np.uint16)
        sig = np.zeros(1200, np.uint8)
        ns, ub = C.c_int(), C.c_int()
        st = self.lib().orc_locator(C.byref(self.c), alg, _ptr(np.ascontiguousarray(S, np.uint8)),
                                    _ptr(er) if len(er) else None, len(er), _ptr(sig), C.byref(ns), C.byref(ub))
        return st, sig[: ns.value].copy(), ub.value

    def correct_hard(self, alg, frames, erasures=()):
        """frames: (B, n) uint8 (symbols) or float32 (sign -> bit).  Returns
        out (B,n) u8, nerr (B,) i32, status (B,) i32, ref_ub (B,) i32."""
        frames = np.ascontiguousarray(frames)
        is_f = frames.dtype == np.float32
        if not is_f:
            frames = frames.astype(np.uint8)
        frames = frames.reshape(-1, self.n)
        B = frames.shape[0]
        out = np.zeros((B, self.n), np.uint8)
        nerr = np.zeros(B, np.int32)
        status = np.zeros(B, np.int32)
        ub = np.zeros(B, np.int32)
        er = np.asarray(erasures, np.uint16)
        fn = self.lib().orc_correct_hard_f32 if is_f else self.lib().orc_correct_hard
        ne, u = C.c_int(), C.c_int()
        for i in range(B):
            status[i] = fn(C.byref(self.c), alg, _ptr(frames[i]), _ptr(er) if len(er) else None, len(er),
                           _ptr(out[i]), C.byref(ne), C.byref(u))
            nerr[i], ub[i] = ne.value, u.value
        return out, nerr, status, ub

    def minsum(self, variant, iterations, y, alpha=1.0, beta=0.0, stop=O2, erasures=(), fast=False):
        """y: (B, n) float32.  Returns b (B,n) u8, L (B,n) f32, iters (B,) u32, status (B,) i32."""
        y = np.ascontiguousarray(y, np.float32).reshape(-1, self.n)
        B = y.shape[0]
        b = np.zeros((B, self.n), np.uint8)
        L = np.zeros((B, self.n), np.float32)
        iters = np.zeros(B, np.uint32)
        status = np.zeros(B, np.int32)
        er = np.asarray(erasures, np.uint16)
        it = C.c_uint()
        lib = self.lib()
        for i in range(B):
            if fast:
                assert len(er) == 0
                status[i] = lib.orc_minsum_fast(C.byref(self.c), variant, iterations, C.c_double(alpha),
                                                C.c_double(beta), stop, _ptr(y[i]), _ptr(b[i]), _ptr(L[i]),
                                                C.byref(it))
            else:
                status[i] = lib.orc_minsum(C.byref(self.c), variant, iterations, C.c_double(alpha),
                                           C.c_double(beta), stop, _ptr(y[i]), _ptr(er) if len(er) else None,
                                           len(er), _ptr(b[i]), _ptr(L[i]), C.byref(it))
            iters[i] = it.value
        return b, L, iters, status


class RefLib:
    """The real reference (oracle/_ref/libccref_o{0,1}.so)."""
    _libs = {}

    @staticmethod
    def available():
        return all(os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libccref_o%d.so" % i)) for i in (0, 1))

    @classmethod
    def get(cls, fixed):
        fixed = int(bool(fixed))
        if fixed not in cls._libs:
            lib = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libccref_o%d.so" % fixed))
            assert lib.ref_fix_end() == fixed
            cls._libs[fixed] = cls(lib)
        return cls._libs[fixed]

    def __init__(self, lib):
        self.lib = lib

    def info(self, cid):
        v = [C.c_int() for _ in range(4)] + [C.c_uint() for _ in range(5)] + [C.c_double()]
        assert self.lib.ref_code_info(cid, *[C.byref(x) for x in v]) == 0
        keys = ("family", "q", "cap_kind", "cap", "n", "k", "l", "t", "dmin", "rate")
        return dict(zip(keys, [x.value for x in v]))

    def poly(self, cid, which):
        out = np.zeros(512, np.uint8)
        n = self.lib.ref_get_poly(cid, which, _ptr(out), 512)
        assert n >= 0
        return out[:n].copy()

    def to_string(self, cid, alg):
        buf = C.create_string_buffer(128)
        self.lib.ref_to_string(cid, alg, buf, 128)
        return buf.value.decode()

    def H(self, cid):
        i = self.info(cid)
        H = np.zeros((i["k"], i["n"]), np.uint8)
        self.lib.ref_get_H(cid, _ptr(H))
        return H

    def encode(self, cid, msg):
        i = self.info(cid)
        msg = np.ascontiguousarray(msg, np.uint8).reshape(-1, i["l"])
        cw = np.zeros((msg.shape[0], i["n"]), np.uint8)
        w = C.create_string_buffer(256)
        for f in range(msg.shape[0]):
            st = self.lib.ref_encode(cid, _ptr(msg[f]), _ptr(cw[f]), w, 256)
            assert st == 0, w.value
        return cw

    def encode_mult(self, cid, msg):
        """::cyclic::encode(g, a, multiplication_tag) zero-padded to n (cyclic.h:29-33, :303-310)."""
        i = self.info(cid)
        msg = np.ascontiguousarray(msg, np.uint8).reshape(-1, i["l"])
        cw = np.zeros((msg.shape[0], i["n"]), np.uint8)
        w = C.create_string_buffer(256)
        for f in range(msg.shape[0]):
            assert self.lib.ref_encode_mult(cid, _ptr(msg[f]), _ptr(cw[f]), w, 256) == 0, w.value
        return cw

    def decode_mult(self, cid, cw):
        """::cyclic::decode(g, b, multiplication_tag) = b / g zero-padded to l (cyclic.h:42-46, :318-325)."""
        i = self.info(cid)
        cw = np.ascontiguousarray(cw, np.uint8).reshape(-1, i["n"])
        msg = np.zeros((cw.shape[0], i["l"]), np.uint8)
        w = C.create_string_buffer(256)
        for f in range(cw.shape[0]):
            assert self.lib.ref_decode_mult(cid, _ptr(cw[f]), _ptr(msg[f]), w, 256) == 0, w.value
        return msg

    def correct(self, cid, alg, frames, erasures=(), decode=False):
        """Returns out (B, n or l) u8, status (B,), messages list."""
        i = self.info(cid)
        frames = np.ascontiguousarray(frames)
        is_f = frames.dtype == np.float32
        if not is_f:
            frames = frames.astype(np.uint8)
        frames = frames.reshape(-1, i["n"])
        B = frames.shape[0]
        width = i["l"] if decode else i["n"]
        out = np.zeros((B, width), np.uint8)
        status = np.zeros(B, np.int32)
        msgs = []
        er = np.asarray(erasures, np.uint32)
        w = C.create_string_buffer(512)
        fn = self.lib.ref_decode_u8 if decode else (self.lib.ref_correct_f32 if is_f else self.lib.ref_correct_u8)
        for f in range(B):
            status[f] = fn(cid, alg, _ptr(frames[f]), _ptr(er) if len(er) else None, len(er), _ptr(out[f]), w, 512)
            msgs.append(w.value.decode(errors="replace"))
        return out, status, msgs

    def locator(self, cid, alg, frame, erasures=()):
        i = self.info(cid)
        frame = np.ascontiguousarray(frame, np.uint8)
        S = np.zeros(2 * i["t"], np.uint8)
        sig = np.zeros(1200, np.uint8)
        ns = C.c_int()
        er = np.asarray(erasures, np.uint32)
        w = C.create_string_buffer(512)
        st = self.lib.ref_locator(cid, alg, _ptr(frame), _ptr(er) if len(er) else None, len(er), _ptr(S), _ptr(sig),
                                  C.byref(ns), 1200, w, 512)
        return st, S, sig[: ns.value].copy(), w.value.decode(errors="replace")

    def minsum(self, cid, variant, iters, utype, y):
        """Returns b, L, iter, status arrays (status 0 ok, 1 decoding_failure)."""
        i = self.info(cid)
        y = np.ascontiguousarray(y, np.float32).reshape(-1, i["n"])
        B = y.shape[0]
        b = np.zeros((B, i["n"]), np.uint8)
        L = np.zeros((B, i["n"]), np.float32)
        it = np.zeros(B, np.uint32)
        st = np.zeros(B, np.int32)
        sec = C.c_double()
        rc = self.lib.ref_minsum_batch(cid, variant, iters, utype, _ptr(y), C.c_size_t(B), _ptr(b), _ptr(L), _ptr(it),
                                       _ptr(st), C.byref(sec))
        assert rc == 0
        self.last_seconds = sec.value
        return b, L, it, st

    def correct_batch_timed(self, cid, alg, frames):
        i = self.info(cid)
        frames = np.ascontiguousarray(frames, np.uint8).reshape(-1, i["n"])
        out = np.zeros_like(frames)
        st = np.zeros(frames.shape[0], np.int32)
        sec = C.c_double()
        rc = self.lib.ref_correct_u8_batch(cid, alg, _ptr(frames), C.c_size_t(frames.shape[0]), _ptr(out), _ptr(st),
                                           C.byref(sec))
        assert rc == 0
        return out, st, sec.value

    def H_alt(self, cid):
        i = self.info(cid)
        H = np.zeros((i["t"] * i["q"], i["n"]), np.uint8)
        assert self.lib.ref_get_H_alt(cid, _ptr(H)) == 0
        return H

    def minsum_alt(self, cid, variant, iters, utype, y):
        """min_sum<float, U>(code.H_alt<U>(), y, tag)"""
        i = self.info(cid)
        y = np.ascontiguousarray(y, np.float32).reshape(-1, i["n"])
        B = y.shape[0]
        b = np.zeros((B, i["n"]), np.uint8)
        L = np.zeros((B, i["n"]), np.float32)
        it = np.zeros(B, np.uint32)
        st = np.zeros(B, np.int32)
        rc = self.lib.ref_minsum_alt_batch(cid, variant, iters, utype, _ptr(y), C.c_size_t(B), _ptr(b), _ptr(L),
                                           _ptr(it), _ptr(st))
        assert rc == 0
        return b, L, it, st

    def soft_class(self, selector, y, erasures=()):
        y = np.ascontiguousarray(y, np.float32)
        out = np.zeros(len(y), np.uint8)
        er = np.asarray(erasures, np.uint32)
        w = C.create_string_buffer(256)
        st = self.lib.ref_soft_class(selector, _ptr(y), _ptr(er) if len(er) else None, len(er), _ptr(out), w, 256)
        return st, out, w.value.decode(errors="replace")

    def tag_constants(self):
        v = [C.c_double() for _ in range(6)]
        self.lib.ref_tag_constants(*[C.byref(x) for x in v])
        return [x.value for x in v]


def awgn_llr(rng, codewords, rate, ebno_db):
    """y = (1 - 2c) + sigma*N(0,1) in float32; sigma per simulation.c++:83-85."""
    sigma = 1.0 / np.sqrt(2.0 * rate * 10.0 ** (ebno_db / 10.0))
    c = np.asarray(codewords, np.float32)
    noise = rng.standard_normal(c.shape).astype(np.float32)
    return ((1.0 - 2.0 * c) + np.float32(sigma) * noise).astype(np.float32)


class RefWide:
    """The real reference on GF(2^q), q > 8 (oracle/ref_driver.cc, "wide" section): uint16 symbols.
    wide id 0 = primitive_bch<9, errors<3>> with modular polynomial 0x211, id 1 = rs<10, errors<4>> with 0x409."""
    CODES = {0: (BCH, 9, 3, 0x211), 1: (RS, 10, 4, 0x409)}

    @staticmethod
    def available():
        return RefLib.available() and hasattr(RefLib.get(1).lib, "refw_num_codes")

    def __init__(self, wid):
        self.lib = RefLib.get(1).lib
        self.wid = wid
        v = [C.c_int()] + [C.c_uint() for _ in range(6)]
        assert self.lib.refw_info(wid, *[C.byref(x) for x in v]) == 0
        self.family, self.q, self.n, self.k, self.l, self.t, self.dmin = [x.value for x in v]

    def poly(self, which):
        out = np.zeros(1 << 16, np.uint16)
        m = self.lib.refw_get_poly(self.wid, which, _ptr(out), out.size)
        assert m >= 0
        return out[:m].copy()

    def to_string(self, alg):
        buf = C.create_string_buffer(128)
        self.lib.refw_to_string(self.wid, alg, buf, 128)
        return buf.value.decode()

    def encode(self, msg):
        msg = np.ascontiguousarray(msg, np.uint16).reshape(-1, self.l)
        cw = np.zeros((msg.shape[0], self.n), np.uint16)
        w = C.create_string_buffer(256)
        for f in range(msg.shape[0]):
            assert self.lib.refw_encode(self.wid, _ptr(msg[f]), _ptr(cw[f]), w, 256) == 0, w.value
        return cw

    def correct(self, alg, frames, erasures=None, decode=False):
        """Returns out (B, n or l) u16, status (B,) (0 ok, 1 decoding_failure, 2 runtime_error), messages."""
        frames = np.ascontiguousarray(frames, np.uint16).reshape(-1, self.n)
        B = frames.shape[0]
        out = np.zeros((B, self.l if decode else self.n), np.uint16)
        status = np.zeros(B, np.int32)
        msgs = []
        w = C.create_string_buffer(512)
        fn = self.lib.refw_decode if decode else self.lib.refw_correct
        for f in range(B):
            er = np.asarray(erasures[f] if erasures is not None else (), np.uint32)
            status[f] = fn(self.wid, alg, _ptr(frames[f]), _ptr(er) if len(er) else None, len(er), _ptr(out[f]), w, 512)
            msgs.append(w.value.decode(errors="replace"))
        return out, status, msgs
