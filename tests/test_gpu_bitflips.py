"""The reference's own statistical known answer: exhaustive bit-flip word-error rates of BCH(31,16,7) for
the nine decoders of src/simulation/bitflips.c++:12-35, as captured from the reference in SURVEY.md App. B.4
(as shipped = stop rule O0, matrix.h:50 repaired = O1; the O1 column reproduces the table printed in
iterative_soft_decoding_of_bch_codes.pdf: MS 29.7 / 79.2 / 96.3 %, SCMS1 29.2 / 76.1 / 96.7 %,
SCMS2 2.4 / 55.1 / 95.3 % for w = 2, 3, 4)."""
import math

import pytest

import channelcoding_amd as cc
from channelcoding_amd import capi
from channelcoding_amd.montecarlo import bitflip_simulation

pytestmark = pytest.mark.gpu

NCK = [math.comb(31, w) for w in range(7)]  # 1, 31, 465, 4495, 31465, 169911, 736281

# SURVEY App. B.4, rows w = 0..6: (O0, O1)
B4 = {
    "MS": [(0, 0), (0.193548, 0), (0.797849, 0.296774), (0.972859, 0.791324), (0.997807, 0.963102),
           (0.999900, 0.994803), (0.999989, 0.996940)],
    "NMS": [(0, 0), (0.354839, 0), (0.946237, 0.073118), (0.999555, 0.681424), (1, 0.987732), (1, 0.999876),
            (1, 0.999997)],
    "OMS": [(0, 0), (0.354839, 0), (0.946237, 0.086022), (0.999555, 0.696774), (1, 0.981885), (1, 0.999706),
            (1, 0.999992)],
    "SCMS1": [(0, 0), (0.193548, 0), (0.797849, 0.292473), (0.972859, 0.760845), (0.997807, 0.966630),
              (0.999900, 0.997740), (0.999989, 0.997766)],
    "SCMS2": [(0, 0), (0.161290, 0), (0.888172, 0.023656), (0.997108, 0.550167), (0.999968, 0.952519),
              (0.999988, 0.998599), (1, 0.999990)],
}
TAGS = {
    "MS": lambda: cc.min_sum_tag(50),
    "NMS": lambda: cc.normalized_min_sum_tag(50, (8, 10)),
    "OMS": lambda: cc.offset_min_sum_tag(50, (1, 100)),
    "SCMS1": lambda: cc.self_correcting_1_min_sum_tag(50),
    "SCMS2": lambda: cc.self_correcting_2_min_sum_tag(50),
}


@pytest.mark.parametrize("name", sorted(B4))
def test_bitflip_wer_table(name):
    for col, rule in ((0, capi.STOP_AS_SHIPPED), (1, capi.STOP_PUBLISHED)):
        code = cc.primitive_bch(5, cc.dmin(7), TAGS[name](), stop_rule=rule)
        res = bitflip_simulation(code, 6)()
        for w, r in enumerate(res):
            assert r["patterns"] == NCK[w]
            want = B4[name][w][col]
            assert abs(r["wer"] - want) < 6e-7, (name, rule, w, r["wer"], want)
    # exact counts stated in the survey for MS / O1
    if name == "MS":
        assert res[2]["word_errors"] == 138 and res[3]["word_errors"] == 3557


def test_bitflip_2dnms_default_equals_ms_and_algebraic():
    """2D-NMS with the default template arguments is plain MS (Q11); BM = PGZ = Euklid: 0 for w <= 3, 1 for w >= 4."""
    ms = bitflip_simulation(cc.primitive_bch(5, cc.dmin(7), cc.min_sum_tag(50), stop_rule=capi.STOP_PUBLISHED), 4)()
    d2 = bitflip_simulation(cc.primitive_bch(5, cc.dmin(7), cc.normalized_2d_min_sum_tag(50),
                                             stop_rule=capi.STOP_PUBLISHED), 4)()
    assert [r["word_errors"] for r in ms] == [r["word_errors"] for r in d2]
    for tag in (cc.berlekamp_massey_tag(), cc.peterson_gorenstein_zierler_tag(), cc.euklid_tag()):
        res = bitflip_simulation(cc.primitive_bch(5, cc.dmin(7), tag), 5)()
        assert [r["word_errors"] for r in res] == [0, 0, 0, 0, NCK[4], NCK[5]]


def test_bitflip_log_format(tmp_path):
    code = cc.primitive_bch(5, cc.dmin(7), cc.berlekamp_massey_tag())
    bitflip_simulation(code, 2, log_dir=str(tmp_path))()
    lines = (tmp_path / "(31, 16, 7)-BM.log").read_text().splitlines()
    assert lines[0] == "%7s %21s" % ("errors", "wer") and lines[1].startswith("      0 0.0")


def test_benchmark_cli_bitflip_and_awgn(tmp_path):
    """python -m channelcoding_amd.benchmark with the options of src/simulation/benchmark.c++:312-372."""
    from channelcoding_amd import benchmark
    rc = benchmark.main(["--simulation", "bitflip", "--algorithm", "MS", "--algorithm", "bm", "--k", "5", "--dmin", "7",
                         "--errors", "3", "--stop-rule", "1", "--log-dir", str(tmp_path)])
    assert rc == 0
    ms = (tmp_path / "(31, 16, 7)-MS.log").read_text().splitlines()
    assert ms[0] == "%7s %21s" % ("errors", "wer")
    assert [float(l.split()[1]) for l in ms[1:]] == pytest.approx([0, 0, 138 / 465, 3557 / 4495], abs=1e-12)
    bm = (tmp_path / "(31, 16, 7)-BM.log").read_text().splitlines()
    assert [float(l.split()[1]) for l in bm[1:]] == [0, 0, 0, 0]
    rc = benchmark.main(["-simulation", "awgn", "-algorithm", "nms", "-k", "6", "-dmin", "7", "-seed", "3",
                         "--max-samples", "20000", "--log-dir", str(tmp_path)])
    assert rc == 0
    rows = (tmp_path / "(63, 45, 7)-NMS.log").read_text().splitlines()
    wer = [float(l.split()[1]) for l in rows[1:]]
    assert rows[0] == "%7s %21s" % ("ebno", "wer") and len(wer) >= 8
    assert wer[0] > 0.1 and wer[-1] < 1e-3 and all(a >= b - 0.02 for a, b in zip(wer, wer[1:]))
    with pytest.raises(RuntimeError):  # "File ... already exists." simulation.c++:72-81
        benchmark.main(["--simulation", "bitflip", "--algorithm", "bm", "--k", "5", "--dmin", "7",
                        "--log-dir", str(tmp_path)])
