"""CPU-side checks of the product library: it loads, exports every symbol the
header declares, builds the same code constants as the reference (golden
constants.json), and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import golden_util as G
from checkers import REF_CODES

import channelcoding_amd as cc
from channelcoding_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "channelcoding_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cc_[A-Za-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(capi.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), s
    assert syms == capi.exported_symbols()  # the ctypes table binds exactly the header
    assert b"gfx950" in capi.lib().cc_version()


def _host_code(family, q, t, alg=capi.ALG_BM, **kw):
    cls = cc.primitive_bch if family == 0 else cc.rs
    tag = {capi.ALG_PGZ: cc.peterson_gorenstein_zierler_tag, capi.ALG_BM: cc.berlekamp_massey_tag,
           capi.ALG_EUKLID: cc.euklid_tag}[alg]
    return cls(q, cc.errors(t), tag(), device=capi.DEVICE_NONE, **kw)


def test_code_constants_match_reference():
    for cid, c in G.constants().items():
        fam, q, t = REF_CODES[cid]
        for alg, name in ((capi.ALG_PGZ, "PGZ"), (capi.ALG_BM, "BM"), (capi.ALG_EUKLID, "EUKLID")):
            code = _host_code(fam, q, t, alg)
            assert code.to_string() == c["to_string"][name]
        assert (code.n, code.k, code.l, code.t, code.dmin) == (c["n"], c["k"], c["l"], c["t"], c["dmin"])
        assert abs(code.rate - c["rate"]) < 1e-15
        assert list(code.g) == c["g"] and list(code.h) == c["h"] and list(code.roots) == c["roots"]
        H = code.H()
        assert [j for j in range(code.n) if H[0, j]] == c["row0_support"]
        for i in range(1, code.k):
            assert np.array_equal(H[i, i:], H[0, : code.n - i]) and not H[i, :i].any()


def test_capability_and_tags():
    assert cc.dmin(7).t == 3 and cc.dmin(6).t == 2 and cc.errors(2).t == 2  # codes.h:19-26
    d2 = cc.normalized_2d_min_sum_tag(10)
    assert (d2.alpha, d2.beta) == (1.0, 1.0)  # Q11: defaults collapse to plain MS
    d2 = cc.normalized_2d_min_sum_tag(10, (3, 4), (9, 10))
    assert (d2.alpha, d2.beta) == (0.75, 2.25)
    assert cc.normalized_min_sum_tag(10, (8, 10)).alpha == 0.8
    assert cc.offset_min_sum_tag(10, (1, 100)).beta == 0.01
    soft = cc.primitive_bch(6, cc.errors(3), cc.min_sum_tag(10), device=capi.DEVICE_NONE)
    assert soft.to_string() == "(63, 45, 7)-MS"
    assert cc.primitive_bch(5, cc.dmin(7), cc.self_correcting_2_min_sum_tag(), device=capi.DEVICE_NONE).to_string() \
        == "(31, 16, 7)-SCMS2"


def test_invalid_descriptors():
    with pytest.raises(cc.CcError):
        cc.primitive_bch(9, cc.errors(2), device=capi.DEVICE_NONE)  # q > 8: no default modular polynomial
    with pytest.raises(cc.CcError):
        cc.primitive_bch(4, cc.errors(8), device=capi.DEVICE_NONE)
    with pytest.raises(cc.CcError) as e:
        cc.rs(4, cc.errors(3), cc.min_sum_tag(10), device=capi.DEVICE_NONE)  # non-binary H
    assert e.value.status == capi.ERR_UNSUPPORTED


def test_no_cpu_fallback():
    """Without a device every compute entry point fails loudly."""
    code = cc.primitive_bch(4, cc.errors(2), cc.min_sum_tag(10), device=capi.DEVICE_NONE)
    y = np.ones((2, 15), np.float32)
    with pytest.raises(cc.CcError) as e:
        code.correct_batch(y)
    assert e.value.status == capi.ERR_NO_DEVICE
    hard = cc.primitive_bch(4, cc.errors(2), cc.berlekamp_massey_tag(), device=capi.DEVICE_NONE)
    for fn, arg in ((hard.correct_batch, np.zeros((1, 15), np.uint8)), (hard.encode_batch, np.zeros((1, 7), np.uint8))):
        with pytest.raises(cc.CcError) as e:
            fn(arg)
        assert e.value.status in (capi.ERR_NO_DEVICE, capi.ERR_UNSUPPORTED)
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(cc.CcError) as e:
            cc.primitive_bch(4, cc.errors(2))  # default device: needs a GPU
        assert e.value.status == capi.ERR_NO_DEVICE


def test_h_alt_and_custom_matrix_arguments():
    """cc_get_H_alt needs no device; cc_code_create_with_H validates like the reference's free min_sum would."""
    for cid in G.ALT_CIDS:
        fam, q, t = REF_CODES[cid]
        code = cc.primitive_bch(q, cc.errors(t), device=capi.DEVICE_NONE)
        H, _, _, _ = G.minsum_alt_cases(cid)
        assert np.array_equal(code.H_alt(), H)
    H = cc.primitive_bch(4, cc.errors(2), device=capi.DEVICE_NONE).H_alt()
    soft = cc.primitive_bch(4, cc.errors(2), cc.min_sum_tag(10), device=capi.DEVICE_NONE, H=H)
    assert soft.n == 15 and soft.to_string() == "(15, 7, 5)-MS"
    with pytest.raises(capi.CcError):  # a matrix only makes sense for the min-sum family
        cc.primitive_bch(4, cc.errors(2), cc.berlekamp_massey_tag(), device=capi.DEVICE_NONE, H=H)
    with pytest.raises(ValueError):  # wrong width
        cc.primitive_bch(4, cc.errors(2), cc.min_sum_tag(10), device=capi.DEVICE_NONE, H=H[:, :14])
    with pytest.raises(capi.CcError):  # non-binary entries
        cc.primitive_bch(4, cc.errors(2), cc.min_sum_tag(10), device=capi.DEVICE_NONE, H=H * 2)


def test_min_sum_decoder_host_logic():
    """cc_minsum_create: the free min_sum(H, y, tag) -- a matrix, no code."""
    rng = np.random.default_rng(5)
    H = (rng.random((20, 100)) < 0.1).astype(np.uint8)
    dec = cc.min_sum_decoder(H, cc.normalized_min_sum_tag(20, 0.8), device=capi.DEVICE_NONE)
    assert (dec.n, dec.k, dec.l) == (100, 20, 80) and dec.to_string() == "20x100-NMS"
    assert np.array_equal(dec.H(), H)
    assert dec.kernel_info()["kernel"] == "minsum_generic_kernel<C=2,W=64>"
    assert cc.min_sum_decoder(H[:, :15], device=capi.DEVICE_NONE).kernel_info()["frames_per_workgroup"] == 4
    for call in (lambda: dec.encode_batch(np.zeros((1, 80), np.uint8)), dec.H_alt,
                 lambda: dec.extract_batch(np.zeros((1, 100), np.uint8))):
        with pytest.raises(capi.CcError):
            call()
    with pytest.raises(capi.CcError):  # no CPU fallback here either
        dec.correct_batch(np.ones((1, 100), np.float32))
    with pytest.raises(capi.CcError):
        cc.min_sum_decoder(np.zeros((4, 2049), np.uint8), device=capi.DEVICE_NONE)  # more than 2048 columns
    with pytest.raises(capi.CcError):
        cc.min_sum_decoder(H, cc.berlekamp_massey_tag(), device=capi.DEVICE_NONE)
    with pytest.raises(capi.CcError):
        cc.min_sum_decoder(H * 3, device=capi.DEVICE_NONE)


def test_benchmark_cli_selection():
    """Decoder registry and option semantics of src/simulation/benchmark.c++ (no GPU needed)."""
    from channelcoding_amd import benchmark
    reg = benchmark.registry()
    assert len(reg) == 108  # 9 algorithms x k in 5..7 x dmin in 3,5,7,9 (benchmark.c++:23-166)
    for name, k, d, shown in reg[::7]:
        code = benchmark.build(name, k, d, capi.STOP_PARITY, device=capi.DEVICE_NONE)
        text = code.to_string()
        assert text.rsplit("-", 1)[1].lower() == name  # the key the reference derives from to_string(), :212-216
        assert text.startswith("(%d, " % ((1 << k) - 1)) and shown == code.dmin >= d
    # --dmin selects on the distance PRINTED by to_string(): primitive_bch<5, dmin<9>> is "(31, 11, 11)"
    assert benchmark.reported_distances() == sorted({r[3] for r in reg}) and 11 in benchmark.reported_distances()
    assert len(benchmark.select(None, None, None)) == 108
    assert benchmark.select(["BM"], ["5"], ["all"]) == [("bm", 5, d) for d in (3, 5, 7, 9)]
    assert benchmark.select(["ms", "nms"], ["5"], ["11"]) == [("ms", 5, 9), ("nms", 5, 9)]
    assert benchmark.select(["ms"], ["5"], ["9"]) == []
    with pytest.raises(SystemExit):
        benchmark.select(["viterbi"], None, None)
    with pytest.raises(SystemExit):
        benchmark.select(None, ["8"], None)
    assert benchmark.main(["--simulation", "fading"]) == 1 and benchmark.main(["--bogus"]) == 1


def test_wide_field_code_constants_match_reference_vectors():
    """GF(2^q), q > 8 (galois.h:23-25,44-53,57-75): g, h, roots, (n, k, l, dmin) and to_string of the codes in
    tests/golden/wide.npz (vectors of the real reference) from the host-side construction, no GPU needed."""
    import os
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wide.npz"))
    tags = (cc.peterson_gorenstein_zierler_tag, cc.berlekamp_massey_tag, cc.euklid_tag)
    for wid in (0, 1):
        p = "w%d_" % wid
        fam, q, t, poly, n, k, l, dmin = [int(v) for v in gold[p + "params"]]
        mk = cc.primitive_bch if fam == 0 else cc.rs
        codes = [mk(q, cc.errors(t), tag(), device=capi.DEVICE_NONE, modular_polynomial=poly) for tag in tags]
        code = codes[0]
        assert code.wide and (code.n, code.k, code.l, code.dmin) == (n, k, l, dmin)
        assert [c.to_string() for c in codes] == list(gold[p + "names"])
        assert np.array_equal(code.g, gold[p + "g"]) and np.array_equal(code.h, gold[p + "h"])
        assert np.array_equal(code.roots, gold[p + "roots"])
        with pytest.raises(cc.CcError) as e:  # no CPU decode path for wide symbols either
            code.encode_batch(np.zeros((1, l), np.uint16))
        assert e.value.status == capi.ERR_NO_DEVICE
    # the reference has no default polynomial beyond q = 8 (galois.h:57-67): the caller must name one
    with pytest.raises(cc.CcError):
        cc.primitive_bch(9, cc.errors(3), cc.berlekamp_massey_tag(), device=capi.DEVICE_NONE)
    with pytest.raises(cc.CcError):
        cc.primitive_bch(9, cc.errors(3), cc.berlekamp_massey_tag(), device=capi.DEVICE_NONE, modular_polynomial=0x201)
    # every q up to 15 (uint16_t storage, galois.h:44-53) constructs; n = 2^q - 1
    for q, poly in ((9, 0x211), (10, 0x409), (11, 0x805), (12, 0x1053), (13, 0x201B), (14, 0x4443), (15, 0x8003)):
        c = cc.rs(q, cc.errors(2), cc.euklid_tag(), device=capi.DEVICE_NONE, modular_polynomial=poly)
        assert c.n == (1 << q) - 1 and c.k == 4 and c.to_string() == "(%d, %d, 6)-EUKLID" % (c.n, c.n - 4)


def test_diagonal_deal_is_a_partition_with_chained_pairs():
    """Host logic of the diagonal min-sum kernel (csrc/minsum_diag.hip): the row-0 support of every code with a
    diagonal geometry is dealt to D slots x LPF lanes exactly once; in a geometry with links, slots 2p and 2p + 1
    of EVERY lane hold adjacent diagonals s and s + 1 (what lets the kernel pass a column from one to the other
    through registers: csrc/minsum_diag_impl.hpp, CHAIN)."""
    import ctypes as C
    lib = capi.lib()
    seen_links = 0
    for q, ts in ((4, (1, 2, 3)), (5, (1, 2, 3, 5)), (6, (1, 2, 3, 4)), (7, (1, 2, 3, 4)), (8, (1, 2, 3, 4))):
        for t in ts:
            code = cc.primitive_bch(q, cc.errors(t), cc.min_sum_tag(5), device=capi.DEVICE_NONE)
            out = np.zeros(1024, np.uint16)
            D, LPF, links = C.c_uint32(), C.c_uint32(), C.c_uint32()
            m = lib.cc_diag_table(code._h, out.ctypes.data_as(C.c_void_p), out.size, C.byref(D), C.byref(LPF),
                                  C.byref(links))
            assert m == D.value * LPF.value and m > 0, (q, t, m)
            table = out[:m].reshape(D.value, LPF.value)
            H = code.H()
            support = [j for j in range(code.n) if H[0, j]]
            dealt = sorted(int(v) for v in table.ravel() if v != 0xFFFF)
            assert dealt == support, (q, t)
            for p in range(links.value):
                assert np.array_equal(table[2 * p + 1].astype(int), table[2 * p].astype(int) + 1), (q, t, p)
            seen_links += links.value
            again = np.zeros(1024, np.uint16)
            lib.cc_diag_table(code._h, again.ctypes.data_as(C.c_void_p), again.size, None, None, None)
            assert np.array_equal(out, again)  # deterministic
    assert seen_links >= 10  # the headline code alone has two
    rs = cc.rs(8, cc.errors(16), cc.berlekamp_massey_tag(), device=capi.DEVICE_NONE)
    assert lib.cc_diag_table(rs._h, out.ctypes.data_as(C.c_void_p), out.size, None, None, None) == 0


def test_dispatched_kernels_do_not_spill():
    """Register metadata of the built kernels (NT_AMDGPU_METADATA notes of the device code objects inside
    csrc/build/*.o, profiles/tools/kernel_meta.py): no dispatched instantiation may spill registers to scratch -- a
    spill is silent and shows only as lost throughput.  Known and documented: the one geometry with 256
    message registers; the bound below is what the compiler does today, so a regression still fails."""
    import glob
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "profiles", "tools"))
    from kernel_meta import kernel_meta
    objs = sorted(glob.glob(os.path.join(root, "channelcoding_amd", "csrc", "build", "*.o")))
    if not objs or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("no built objects / LLVM tools here")
    # (object, variant id): spilled VGPRs the compiler produces today.  BCH(255,223) (K = 32, 256 message registers, one
    # wave per SIMD) parks two registers in every variant; the self-correcting variants no longer spill anywhere (they
    # keep q instead of r since round 3, profiles/r03_experiments.md E38)
    allowed = {("geo_g255_32", v): 2 for v in (16, 17, 18, 19, 20, 21)}
    allowed[("geo_g63_24", 19)] = 9  # SCMS1 with bit words at three waves per SIMD: faster than the q-only form all the same (E38)
    # BCH(127,99), 196 message registers: two waves per SIMD with these spills beat one wave without (MS 43.9 -> 53.1 M, E41)
    allowed.update({("geo_g127_28", v): 38 for v in (16, 17, 18, 19, 20, 21)})
    seen = 0
    for path in objs:
        name = __import__("re").sub(r"(_p\d+)?\.o$", "", os.path.basename(path))  # geo_NAME_p<part>.o -> geo_NAME
        for k in kernel_meta(path):
            seen += 1
            spill = k.get("vgpr_spill_count", 0)
            m = __import__("re").search(r"minsum_diag_kernel<\d+, \d+, (\d+),", k["demangled"])
            limit = allowed.get((name, int(m.group(1))), 0) if m else 0
            assert spill <= limit, (name, k["demangled"][:120], spill)
    assert seen > 50
