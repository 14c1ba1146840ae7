"""GPU tests of the device Monte-Carlo path: cc_awgn_llr_dev (Philox + Box-Muller channel) and
cc_mc_run_dev (channel -> decode -> count).  The counters must equal a host-side count over the very
same frames decoded through the plain batch API, and must not depend on sharding or chunking."""
import ctypes as C

import numpy as np
import pytest

import channelcoding_amd as cc
from channelcoding_amd import capi
from channelcoding_amd.montecarlo import DeviceBackend, awgn_simulation, shard

pytestmark = pytest.mark.gpu


def awgn(code, ebno, seed, first, frames, random_cw):
    import torch
    llr = torch.empty((frames, code.n), dtype=torch.float32, device="cuda")
    sent = torch.empty((frames, code.n), dtype=torch.uint8, device="cuda")
    rc = capi.lib().cc_awgn_llr_dev(code._h, float(ebno), seed, first, frames, int(random_cw),
                                    C.c_void_p(llr.data_ptr()), C.c_void_p(sent.data_ptr()), None)
    capi.check(rc, "cc_awgn_llr_dev")
    torch.cuda.synchronize()
    return llr, sent


def mc(code, ebno, seed, first, frames, random_cw):
    return DeviceBackend(code, random_cw).run(ebno, seed, first, frames).cpu().numpy()


def test_channel_statistics_and_determinism():
    import torch
    code = cc.primitive_bch(8, cc.errors(3), cc.min_sum_tag(20))
    llr, sent = awgn(code, 4.0, 7, 0, 20000, False)
    sigma = code.sigma(4.0)
    assert abs(sigma - 0.468758) < 1e-6
    assert int(sent.sum()) == 0
    assert abs(float(llr.mean()) - 1.0) < 3e-3 and abs(float(llr.std()) - sigma) < 3e-3
    z = ((llr - 1.0) / sigma).flatten()
    assert abs(float((z ** 3).mean())) < 0.02 and abs(float((z ** 4).mean()) - 3.0) < 0.05  # skewness, kurtosis
    # frame noise depends only on (seed, global frame index)
    a, _ = awgn(code, 4.0, 7, 0, 3000, False)
    b1, _ = awgn(code, 4.0, 7, 0, 1234, False)
    b2, _ = awgn(code, 4.0, 7, 1234, 3000 - 1234, False)
    assert torch.equal(a, torch.cat([b1, b2]))
    c, _ = awgn(code, 4.0, 8, 0, 100, False)
    assert not torch.equal(a[:100], c)
    # random codewords: transmitted words are codewords, y = (1 - 2c) + noise with the same noise
    llr2, sent2 = awgn(code, 4.0, 7, 0, 3000, True)
    hard = cc.primitive_bch(8, cc.errors(3), cc.berlekamp_massey_tag())
    chk = hard.correct_batch(sent2)
    assert int((chk["status"] != 0).sum()) == 0 and int(chk["nerr"].sum()) == 0
    assert 0.45 < float(sent2.float().mean()) < 0.55
    assert torch.allclose(llr2 - (1.0 - 2.0 * sent2.float()), a - 1.0, atol=1e-6)


@pytest.mark.parametrize("tag,random_cw", [(cc.min_sum_tag(20), False), (cc.min_sum_tag(20), True),
                                           (cc.normalized_min_sum_tag(10, (8, 10)), True),
                                           (cc.berlekamp_massey_tag(), True), (cc.euklid_tag(), False)])
def test_mc_counters_match_host_count(tag, random_cw):
    code = cc.primitive_bch(8, cc.errors(3), tag)
    frames, seed, first, ebno = 70000, 3, 5000, 5.0  # > one 2^16 chunk
    c = mc(code, ebno, seed, first, frames, random_cw)
    llr, sent = awgn(code, ebno, seed, first, frames, random_cw)
    res = code.correct_batch(llr)
    out, st = res["out"], res["status"]
    biterr = (out != sent).sum(dim=1)
    failed = st != 0
    assert c[capi.MC_FRAMES] == frames
    assert c[capi.MC_BIT_ERRORS] == int(biterr.sum())
    assert c[capi.MC_FAILURES] == int(failed.sum())
    assert c[capi.MC_WORD_ERRORS] == int((failed | (biterr > 0)).sum())
    assert c[capi.MC_UNDETECTED] == int((~failed & (biterr > 0)).sum())
    assert c[capi.MC_CHANNEL_BIT_ERRORS] == int(((llr < 0) != (sent != 0)).sum())
    if tag.soft:
        it = res["iters"].to(int)
        run = it + 1
        run[failed] = tag.iterations
        assert c[capi.MC_ITER_SUM] == int(run.sum())
        hist = np.bincount(it[~failed].cpu().numpy(), minlength=56)[:56]
        assert np.array_equal(c[capi.MC_ITER_HIST:capi.MC_ITER_HIST + 56], hist)
    # sharding independence: 1 rank == sum over 3 ranks
    total = np.zeros_like(c)
    for r in range(3):
        lo, cnt = shard(frames, r, 3)
        total += mc(code, ebno, seed, first + lo, cnt, random_cw)
    assert np.array_equal(total, c)


def test_awgn_simulation_ladder(tmp_path):
    """The harness end to end on one GPU: ladder, adaptive samples, reference-format log."""
    code = cc.primitive_bch(5, cc.dmin(7), cc.berlekamp_massey_tag())
    sim = awgn_simulation(code, step=0.5, seed=0, log_dir=str(tmp_path), max_samples=20000)
    assert abs(sim.start - 1.0) < 1e-9  # the (31,16,7) ladder starts at 1.0 dB
    res = sim()
    assert [r["ebno"] for r in res][:3] == [1.0, 1.5, 2.0] and abs(res[-1]["ebno"] - 8.0) < 1e-9
    wer = [r["wer"] for r in res]
    assert all(a >= b - 0.02 for a, b in zip(wer, wer[1:]))  # monotone up to sampling noise
    # App. B.5 reference values (statistical): 0.5682 at 1.0 dB, 0.2079 at 3.0 dB
    assert abs(wer[0] - 0.5682) < 0.03 and abs(wer[4] - 0.2079) < 0.03
    lines = (tmp_path / "(31, 16, 7)-BM.log").read_text().splitlines()
    assert lines[0] == "%7s %21s" % ("ebno", "wer") and len(lines) == len(res) + 1
    assert lines[1].startswith("      1 ") and "e-01" in lines[1]
    with pytest.raises(RuntimeError):
        awgn_simulation(code, log_dir=str(tmp_path), max_samples=100)()  # refuses to overwrite


def test_mc_unsupported():
    with pytest.raises(cc.CcError) as e:
        mc(cc.rs(4, cc.errors(3), cc.berlekamp_massey_tag()), 4.0, 0, 0, 10, False)
    assert e.value.status == capi.ERR_UNSUPPORTED
