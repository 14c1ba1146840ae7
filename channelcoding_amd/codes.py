"""Host-side mirror of the reference's code classes on top of the C ABI.

Names and argument meaning follow the reference (file:line relative to its repo):

* ``errors`` / ``dmin``                     src/codes/codes.h:7-26
* algorithm tags                             src/codes/hard_decision.h:15-24, src/codes/soft_decision.h:20-73
* ``primitive_bch`` / ``rs``                 src/codes/bch.h:16-19, src/codes/rs.h:6-10
* ``encode`` / ``decode`` / ``correct``      src/codes/cyclic.h:289-344
* ``H`` / ``to_string`` / ``rate`` / ``n`` / ``t``   src/codes/cyclic.h:94-95,:111,:282-287,:346-359
* ``decoding_failure``                       src/codes/codes.h:28-36

Input-sequence convention (cyclic.h:163-184, :220-222): a *signed* element type
(float / signed-int numpy arrays or torch tensors) is a soft value, bit = (x < 0);
an *unsigned* one (uint8 / bool arrays, or plain lists of non-negative ints) holds
symbols.  All decoding runs on the GPU through libchannelcoding_amd.so; there is
no CPU path here.
"""
import ctypes as C
from fractions import Fraction

import numpy as np

from . import _capi as capi
from ._capi import CcError


class decoding_failure(RuntimeError):
    """codes.h:28-36"""


class errors:  # codes.h:7-9
    def __init__(self, e):
        self.value = int(e)
        self.t = int(e)


class dmin:  # codes.h:11-13, correction_capability :19-21
    def __init__(self, d):
        self.value = int(d)
        self.t = (int(d) - 1) // 2


def _ratio(r):
    if isinstance(r, Fraction):
        return r.numerator, r.denominator
    if isinstance(r, (tuple, list)):
        return int(r[0]), int(r[1])
    f = Fraction(r).limit_denominator(1 << 20)
    return f.numerator, f.denominator


class _tag:
    soft = False
    iterations = 0
    alpha = 1.0
    beta = 0.0


class peterson_gorenstein_zierler_tag(_tag):
    alg = capi.ALG_PGZ


class berlekamp_massey_tag(_tag):
    alg = capi.ALG_BM


class euklid_tag(_tag):
    alg = capi.ALG_EUKLID


class min_sum_tag(_tag):  # soft_decision.h:20-23
    alg, soft = capi.ALG_MS, True

    def __init__(self, iterations=50):
        self.iterations = int(iterations)


class normalized_min_sum_tag(_tag):  # :36-42  alpha = num/den
    alg, soft = capi.ALG_NMS, True

    def __init__(self, iterations, ratio=(1, 1)):
        num, den = _ratio(ratio)
        self.iterations, self.alpha = int(iterations), num / den


class offset_min_sum_tag(_tag):  # :44-50  beta = num/den
    alg, soft = capi.ALG_OMS, True

    def __init__(self, iterations=50, ratio=(0, 1)):
        num, den = _ratio(ratio)
        self.iterations, self.beta = int(iterations), num / den


class self_correcting_1_min_sum_tag(_tag):  # :52-56
    alg, soft = capi.ALG_SCMS1, True

    def __init__(self, iterations=50):
        self.iterations = int(iterations)


class self_correcting_2_min_sum_tag(_tag):  # :58-62
    alg, soft = capi.ALG_SCMS2, True

    def __init__(self, iterations=50):
        self.iterations = int(iterations)


class normalized_2d_min_sum_tag(_tag):  # :64-73
    alg, soft = capi.ALG_2DNMS, True

    def __init__(self, iterations=50, alpha=(1, 1), beta=(1, 10)):
        an, ad = _ratio(alpha)
        bn, _bd = _ratio(beta)
        self.iterations = int(iterations)
        self.alpha = an / ad
        # soft_decision.h:71 divides Beta::num by Alpha::den (sic): the defaults give alpha = beta = 1
        self.beta = bn / ad


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _ptr(a):
    if a is None:
        return None
    if _is_torch(a):
        return C.c_void_p(a.data_ptr())
    return a.ctypes.data_as(C.c_void_p)


def _stream_handle(t):
    import torch
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _erasure_csr(erasures, B, n):
    """None | flat list (one frame, or shared by every frame) | list of per-frame lists -> (values, offsets)."""
    if erasures is None:
        return None, None
    erasures = list(erasures)
    if len(erasures) and isinstance(erasures[0], (list, tuple, np.ndarray)):
        per = [list(map(int, e)) for e in erasures]
        if len(per) != B:
            raise ValueError("need one erasure list per frame")
    else:
        if not len(erasures):
            return None, None
        per = [list(map(int, erasures))] * B
    off = np.zeros(B + 1, np.uint32)
    off[1:] = np.cumsum([len(p) for p in per])
    vals = np.array([e for p in per for e in p], np.uint16)
    if len(vals) and (vals >= n).any():
        raise IndexError("erasure position out of range")  # copy.at(erasure), cyclic.h:261
    return vals, off


class cyclic:
    """Common part of primitive_bch and rs (cyclic::cyclic<...>, cyclic.h:67-386)."""
    family = None

    def __init__(self, q, capability, algorithm=None, coding="division", mu=1, step=1,
                 stop_rule=capi.STOP_PARITY, device=None, H=None, modular_polynomial=None):
        """modular_polynomial: math::modular_polynomial<> (galois.h:23-25), bit i = coefficient of x^i; None = the
        default of galois.h:18-20, which exists for q <= 8 only.  q > 8: symbols are numpy uint16 / torch int16."""
        if isinstance(capability, int):
            capability = errors(capability)
        algorithm = algorithm if algorithm is not None else peterson_gorenstein_zierler_tag()
        if isinstance(algorithm, type):
            algorithm = algorithm()
        self.algorithm = algorithm
        self.capability = capability
        lib = capi.lib()
        d = capi.Desc()
        lib.cc_desc_init(C.byref(d))
        d.family, d.q, d.t = self.family, int(q), int(capability.t)
        d.mu, d.step = int(mu), int(step)
        d.coding = capi.CODING_MULTIPLICATION if coding in ("multiplication", capi.CODING_MULTIPLICATION) \
            else capi.CODING_DIVISION
        d.algorithm = algorithm.alg
        d.iterations = algorithm.iterations
        d.alpha, d.beta = float(algorithm.alpha), float(algorithm.beta)
        d.stop_rule = int(stop_rule)
        d.device = capi.DEVICE_CURRENT if device is None else int(device)
        d.modular_polynomial = int(modular_polynomial or 0)
        self.wide = int(q) > 8
        self._desc = d
        h = C.c_void_p()
        if H is None:
            capi.check(lib.cc_code_create(C.byref(d), C.byref(h)), "cc_code_create")
        else:  # min_sum<float, U>(matrix, y, tag) on a caller-supplied parity-check matrix, e.g. H_alt()
            Hm = np.ascontiguousarray(H, np.uint8)
            if Hm.ndim != 2 or Hm.shape[1] != (1 << int(q)) - 1:
                raise ValueError("H must be a (rows, n) matrix")
            capi.check(lib.cc_code_create_with_H(C.byref(d), _ptr(Hm), Hm.shape[0], C.byref(h)),
                       "cc_code_create_with_H")
        self._h = h
        self.q = int(q)
        self.n, self.k, self.l = lib.cc_n(h), lib.cc_k(h), lib.cc_l(h)
        self.t, self.dmin, self.rate = lib.cc_t(h), lib.cc_dmin(h), lib.cc_rate(h)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                capi.lib().cc_code_destroy(h)
            except Exception:
                pass

    # ---- introspection ----
    def to_string(self):
        buf = C.create_string_buffer(128)
        capi.check(capi.lib().cc_to_string(self._h, buf, 128), "cc_to_string")
        return buf.value.decode()

    def _poly(self, which):
        if self.wide:
            out = np.zeros(1 << 16, np.uint16)
            m = capi.lib().cc_get_poly_u16(self._h, which, _ptr(out), out.size)
            if m < 0:
                raise CcError(capi.ERR_INVALID_ARGUMENT, "cc_get_poly_u16")
            return out[:m].copy()
        out = np.zeros(512, np.uint8)
        m = capi.lib().cc_get_poly(self._h, which, _ptr(out), 512)
        if m < 0:
            raise CcError(capi.ERR_INVALID_ARGUMENT, "cc_get_poly")
        return out[:m].copy()

    @property
    def g(self):
        return self._poly(0)

    @property
    def h(self):
        return self._poly(1)

    @property
    def roots(self):
        return self._poly(2)

    def H(self):
        H = np.zeros((self.k, self.n), np.uint8)
        capi.check(capi.lib().cc_get_H(self._h, _ptr(H)), "cc_get_H")
        return H

    def kernel_info(self):
        """Which device kernel cc_correct_*_batch dispatches to for this handle (diagnostics, bench.py)."""
        name = C.create_string_buffer(160)
        fpw, thr, lds = C.c_uint32(), C.c_uint32(), C.c_uint32()
        capi.check(capi.lib().cc_kernel_info(self._h, name, 160, C.byref(fpw), C.byref(thr), C.byref(lds)),
                   "cc_kernel_info")
        return {"kernel": name.value.decode(), "frames_per_workgroup": fpw.value, "threads": thr.value,
                "lds_bytes": lds.value}

    def H_alt(self):
        """cyclic::H_alt<uint8_t>() (cyclic.h:361-385), t*q rows."""
        H = np.zeros((self.t * self.q, self.n), np.uint8)
        rows = C.c_uint32()
        capi.check(capi.lib().cc_get_H_alt(self._h, _ptr(H), C.byref(rows)), "cc_get_H_alt")
        return H[: rows.value]

    def sigma(self, ebno_db):
        return capi.lib().cc_sigma(self._h, float(ebno_db))

    # ---- batch API (numpy host arrays or torch CUDA tensors) ----
    # ---- q > 8: 16-bit symbols (numpy uint16 on the host, torch int16 / uint16 on the device) ----
    def _wide_map(self, x, width_in, width_out, host_fn, dev_fn):
        lib = capi.lib()
        if _is_torch(x):
            import torch
            x = x.contiguous()
            if x.element_size() != 2 or x.shape[-1] != width_in:
                raise CcError(capi.ERR_LENGTH, dev_fn)
            B = x.numel() // width_in
            out = torch.empty((B, width_out), dtype=x.dtype, device=x.device)
            capi.check(getattr(lib, dev_fn)(self._h, _ptr(x), _ptr(out), B, _stream_handle(x)), dev_fn)
            return out
        x = np.ascontiguousarray(x, np.uint16)
        if x.shape[-1] != width_in:
            raise CcError(capi.ERR_LENGTH, host_fn)
        x = x.reshape(-1, width_in)
        out = np.zeros((x.shape[0], width_out), np.uint16)
        capi.check(getattr(lib, host_fn)(self._h, _ptr(x), _ptr(out), x.shape[0]), host_fn)
        return out

    def _wide_correct(self, b, erasures):
        lib = capi.lib()
        if _is_torch(b):
            import torch
            b = b.contiguous()
            if b.element_size() != 2 or b.shape[-1] != self.n:
                raise CcError(capi.ERR_LENGTH, "correct_batch")
            B = b.numel() // self.n
            er = off = None
            if erasures is not None:
                ev, eo = _erasure_csr(erasures, B, self.n)
                if ev is not None:
                    ev = ev if ev.size else np.zeros(1, ev.dtype)
                    er = torch.from_numpy(ev.astype(np.int16)).to(b.device)
                    off = torch.from_numpy(eo.astype(np.int32)).to(b.device)
            out = torch.empty((B, self.n), dtype=b.dtype, device=b.device)
            nerr = torch.empty(B, dtype=torch.int32, device=b.device)
            status = torch.empty(B, dtype=torch.int32, device=b.device)
            capi.check(lib.cc_correct_hard_batch_u16_dev(self._h, _ptr(b), _ptr(er), _ptr(off), _ptr(out), _ptr(nerr),
                                                         _ptr(status), B, _stream_handle(b)),
                       "cc_correct_hard_batch_u16_dev")
            return dict(out=out, status=status, nerr=nerr)
        b = np.ascontiguousarray(b, np.uint16)
        if b.shape[-1] != self.n:
            raise CcError(capi.ERR_LENGTH, "correct_batch")
        b = b.reshape(-1, self.n)
        B = b.shape[0]
        er, off = _erasure_csr(erasures, B, self.n)
        out = np.zeros((B, self.n), np.uint16)
        nerr = np.zeros(B, np.int32)
        status = np.zeros(B, np.int32)
        capi.check(lib.cc_correct_hard_batch_u16(self._h, _ptr(b), _ptr(er), _ptr(off), _ptr(out), _ptr(nerr),
                                                 _ptr(status), B), "cc_correct_hard_batch_u16")
        return dict(out=out, status=status, nerr=nerr)

    def encode_batch(self, msg):
        lib = capi.lib()
        if self.wide:
            return self._wide_map(msg, self.l, self.n, "cc_encode_batch_u16", "cc_encode_batch_u16_dev")
        if _is_torch(msg):
            import torch
            msg = msg.contiguous()
            if msg.dtype != torch.uint8 or msg.shape[-1] != self.l:
                raise CcError(capi.ERR_LENGTH, "encode_batch")
            B = msg.numel() // self.l
            cw = torch.empty((B, self.n), dtype=torch.uint8, device=msg.device)
            capi.check(lib.cc_encode_batch_dev(self._h, _ptr(msg), _ptr(cw), B, _stream_handle(msg)),
                       "cc_encode_batch_dev")
            return cw
        msg = np.ascontiguousarray(msg, np.uint8)
        if msg.shape[-1] != self.l:
            raise CcError(capi.ERR_LENGTH, "encode_batch")  # cyclic.h:291-296
        msg = msg.reshape(-1, self.l)
        cw = np.zeros((msg.shape[0], self.n), np.uint8)
        capi.check(lib.cc_encode_batch(self._h, _ptr(msg), _ptr(cw), msg.shape[0]), "cc_encode_batch")
        return cw

    def extract_batch(self, cw):
        lib = capi.lib()
        if self.wide:
            return self._wide_map(cw, self.n, self.l, "cc_extract_batch_u16", "cc_extract_batch_u16_dev")
        if _is_torch(cw):
            import torch
            cw = cw.contiguous()
            B = cw.numel() // self.n
            msg = torch.empty((B, self.l), dtype=torch.uint8, device=cw.device)
            capi.check(lib.cc_extract_batch_dev(self._h, _ptr(cw), _ptr(msg), B, _stream_handle(cw)),
                       "cc_extract_batch_dev")
            return msg
        cw = np.ascontiguousarray(cw, np.uint8).reshape(-1, self.n)
        msg = np.zeros((cw.shape[0], self.l), np.uint8)
        capi.check(lib.cc_extract_batch(self._h, _ptr(cw), _ptr(msg), cw.shape[0]), "cc_extract_batch")
        return msg

    def correct_batch(self, b, erasures=None, want_L=False):
        """Returns a dict: out (B,n) u8, status (B,) i32, and nerr (hard) or iters [+ L] (soft)."""
        lib = capi.lib()
        soft_alg = self.algorithm.soft
        if self.wide and not soft_alg:  # (min-sum takes LLRs and returns bits whatever the symbol width)
            return self._wide_correct(b, erasures)
        if _is_torch(b):
            return self._correct_batch_torch(b, erasures, want_L)
        b = np.asarray(b)
        if b.shape[-1] != self.n:
            raise CcError(capi.ERR_LENGTH, "correct_batch")  # cyclic.h:213-218
        signed = b.dtype.kind in "fi"
        B = b.size // self.n
        er, off = _erasure_csr(erasures, B, self.n)
        out = np.zeros((B, self.n), np.uint8)
        status = np.zeros(B, np.int32)
        if soft_alg:
            if not signed:
                raise TypeError("min-sum needs a signed (soft) input sequence")
            y = np.ascontiguousarray(b, np.float32).reshape(B, self.n)
            L = np.zeros((B, self.n), np.float32) if want_L else None
            iters = np.zeros(B, np.uint16)
            capi.check(lib.cc_correct_soft_batch(self._h, _ptr(y), _ptr(er), _ptr(off), _ptr(out), _ptr(L),
                                                 _ptr(iters), _ptr(status), B), "cc_correct_soft_batch")
            res = dict(out=out, status=status, iters=iters)
            if want_L:
                res["L"] = L
            return res
        nerr = np.zeros(B, np.int32)
        if signed:
            y = np.ascontiguousarray(b, np.float32).reshape(B, self.n)
            capi.check(lib.cc_correct_hard_f32_batch(self._h, _ptr(y), _ptr(er), _ptr(off), _ptr(out), _ptr(nerr),
                                                     _ptr(status), B), "cc_correct_hard_f32_batch")
        else:
            sym = np.ascontiguousarray(b, np.uint8).reshape(B, self.n)
            capi.check(lib.cc_correct_hard_batch(self._h, _ptr(sym), _ptr(er), _ptr(off), _ptr(out), _ptr(nerr),
                                                 _ptr(status), B), "cc_correct_hard_batch")
        return dict(out=out, status=status, nerr=nerr)

    def _correct_batch_torch(self, b, erasures, want_L):
        import torch
        lib = capi.lib()
        b = b.contiguous()
        if b.shape[-1] != self.n:
            raise CcError(capi.ERR_LENGTH, "correct_batch")
        B = b.numel() // self.n
        dev = b.device
        er = off = None
        if erasures is not None:
            ev, eo = _erasure_csr(erasures, B, self.n)
            if ev is not None:
                if ev.size == 0:  # an empty tensor has a null data pointer: keep the list addressable
                    ev = np.zeros(1, ev.dtype)
                er = torch.from_numpy(ev.astype(np.int16)).to(dev)
                off = torch.from_numpy(eo.astype(np.int32)).to(dev)
        out = torch.empty((B, self.n), dtype=torch.uint8, device=dev)
        status = torch.empty(B, dtype=torch.int32, device=dev)
        st = _stream_handle(b)
        if self.algorithm.soft:
            if b.dtype != torch.float32:
                raise TypeError("min-sum needs float32 LLRs")
            L = torch.empty((B, self.n), dtype=torch.float32, device=dev) if want_L else None
            iters = torch.empty(B, dtype=torch.int16, device=dev)
            capi.check(lib.cc_correct_soft_batch_dev(self._h, _ptr(b), _ptr(er), _ptr(off), _ptr(out), _ptr(L),
                                                     _ptr(iters), _ptr(status), B, st), "cc_correct_soft_batch_dev")
            res = dict(out=out, status=status, iters=iters)
            if want_L:
                res["L"] = L
            return res
        nerr = torch.empty(B, dtype=torch.int32, device=dev)
        if b.dtype == torch.float32:
            capi.check(lib.cc_correct_hard_f32_batch_dev(self._h, _ptr(b), _ptr(er), _ptr(off), _ptr(out), _ptr(nerr),
                                                         _ptr(status), B, st), "cc_correct_hard_f32_batch_dev")
        elif b.dtype == torch.uint8:
            capi.check(lib.cc_correct_hard_batch_dev(self._h, _ptr(b), _ptr(er), _ptr(off), _ptr(out), _ptr(nerr),
                                                     _ptr(status), B, st), "cc_correct_hard_batch_dev")
        else:
            raise TypeError("hard decoding takes uint8 symbols or float32 soft values")
        return dict(out=out, status=status, nerr=nerr)

    def decode_batch(self, b, erasures=None):
        """decode = correct + message extraction (cyclic.h:313-327); host arrays go through cc_decode_*_batch."""
        if _is_torch(b):
            res = self.correct_batch(b, erasures)
            res["msg"] = self.extract_batch(res["out"])
            return res
        lib = capi.lib()
        b = np.asarray(b)
        if b.shape[-1] != self.n:
            raise CcError(capi.ERR_LENGTH, "decode_batch")
        B = b.size // self.n
        er, off = _erasure_csr(erasures, B, self.n)
        msg = np.zeros((B, self.l), np.uint8)
        out = np.zeros((B, self.n), np.uint8)
        status = np.zeros(B, np.int32)
        if b.dtype.kind in "fi":
            y = np.ascontiguousarray(b, np.float32).reshape(B, self.n)
            aux = np.zeros(B, np.uint16 if self.algorithm.soft else np.int32)
            if self.algorithm.soft:
                capi.check(lib.cc_decode_soft_batch(self._h, _ptr(y), _ptr(er), _ptr(off), _ptr(msg), _ptr(out),
                                                    _ptr(aux), _ptr(status), B), "cc_decode_soft_batch")
                return dict(out=out, msg=msg, status=status, iters=aux)
            res = self.correct_batch(y, erasures)  # a hard algorithm on channel values: bit = (x < 0), cyclic.h:163-173
            res["msg"] = self.extract_batch(res["out"])
            return res
        sym = np.ascontiguousarray(b, np.uint8).reshape(B, self.n)
        nerr = np.zeros(B, np.int32)
        capi.check(lib.cc_decode_hard_batch(self._h, _ptr(sym), _ptr(er), _ptr(off), _ptr(msg), _ptr(out), _ptr(nerr),
                                            _ptr(status), B), "cc_decode_hard_batch")
        return dict(out=out, msg=msg, status=status, nerr=nerr)

    # ---- single-frame API with the reference's exception behaviour ----
    _MESSAGES = {
        capi.FRAME_NOT_CONVERGED: "Decoding failure",
        capi.FRAME_LOCATOR: "Sigma(x) does not have as many distinct zeroes as its degree",
        capi.FRAME_RECHECK: "Corrected word is not a codeword",
        capi.FRAME_ERASURES: "Number of erasures exceed error correction capability.",
    }

    def _as_frame(self, seq, length, what):
        if isinstance(seq, (list, tuple)):
            arr = np.asarray(seq)
            if arr.dtype.kind == "i" and (arr >= 0).all():
                arr = arr.astype(np.uint8) if (arr < 256).all() else arr
        else:
            arr = np.asarray(seq)
        if arr.ndim != 1 or arr.shape[0] != length:
            raise RuntimeError("%s has the wrong size (%d). Expected %d" % (what, arr.shape[0] if arr.ndim else 0,
                                                                           length))
        return arr

    def encode(self, a):
        a = self._as_frame(a, self.l, "Source code word")
        return self.encode_batch(a.astype(np.uint8)[None, :])[0]

    def correct(self, b, erasures=None):
        b = self._as_frame(b, self.n, "Channel code word")
        res = self.correct_batch(b[None, :], None if not erasures else list(erasures))
        st = int(res["status"][0])
        if st != capi.FRAME_OK:
            raise decoding_failure(self._MESSAGES.get(st, "decoding failure %d" % st))
        return res["out"][0]

    def decode(self, b, erasures=None):
        return self.extract_batch(self.correct(b, erasures)[None, :])[0]


class primitive_bch(cyclic):
    """cyclic::primitive_bch<q, Capability, Sigma, N, Coding> (bch.h:16-19)."""
    family = capi.FAMILY_BCH

    def __init__(self, q, capability, algorithm=None, coding="division", **kw):
        super().__init__(q, capability, algorithm, coding, 1, 1, **kw)


class rs(cyclic):
    """cyclic::rs<q, Capability, Sigma, N, Coding, mu, step> (rs.h:6-10)."""
    family = capi.FAMILY_RS

    def __init__(self, q, capability, algorithm=None, coding="division", mu=1, step=1, **kw):
        super().__init__(q, capability, algorithm, coding, mu, step, **kw)


class min_sum_decoder(cyclic):
    """The free functions min_sum<R, U>(H, y, tag) of soft_decision.h:220-295 bound to one parity-check matrix
    (any rows x cols 0/1 matrix, cols <= 2048; cc_minsum_create).  correct_batch(), correct(), H(), to_string()
    and kernel_info() work; there is no code to encode with."""

    def __init__(self, H, algorithm=None, stop_rule=capi.STOP_PARITY, device=None):
        algorithm = algorithm if algorithm is not None else min_sum_tag()
        if isinstance(algorithm, type):
            algorithm = algorithm()
        Hm = np.ascontiguousarray(H, np.uint8)
        if Hm.ndim != 2:
            raise ValueError("H must be a (rows, cols) matrix")
        self.algorithm = algorithm
        self.capability = None
        lib = capi.lib()
        d = capi.Desc()
        lib.cc_desc_init(C.byref(d))
        d.algorithm = algorithm.alg
        d.iterations = algorithm.iterations
        d.alpha, d.beta = float(algorithm.alpha), float(algorithm.beta)
        d.stop_rule = int(stop_rule)
        d.device = capi.DEVICE_CURRENT if device is None else int(device)
        self.wide = False
        self._desc = d
        h = C.c_void_p()
        capi.check(lib.cc_minsum_create(C.byref(d), _ptr(Hm), Hm.shape[0], Hm.shape[1], C.byref(h)),
                   "cc_minsum_create")
        self._h = h
        self.q = 0
        self.n, self.k, self.l = lib.cc_n(h), lib.cc_k(h), lib.cc_l(h)
        self.t, self.dmin, self.rate = 0, 0, lib.cc_rate(h)


def min_sum(H, y, tag=None, stop_rule=capi.STOP_PARITY):
    """min_sum<float, uint8_t>(H, y, tag): returns (b, L, iteration) like the reference's tuple and raises
    decoding_failure when no iteration satisfies the stop rule (soft_decision.h:199-201).  One frame per call;
    build a min_sum_decoder and use correct_batch for throughput."""
    dec = min_sum_decoder(H, tag, stop_rule)
    y = np.asarray(y, np.float32)
    if y.ndim != 1 or y.shape[0] != dec.n:
        raise CcError(capi.ERR_LENGTH, "min_sum")
    res = dec.correct_batch(y[None, :], want_L=True)
    if int(res["status"][0]) != capi.FRAME_OK:
        raise decoding_failure(cyclic._MESSAGES[capi.FRAME_NOT_CONVERGED])
    return res["out"][0], res["L"][0], int(res["iters"][0])
