"""GPU parity tests of two-pass min-sum decoding (csrc/minsum_diag.hip: launch_two_pass): the message-free first pass
with the general kernel on the compacted rest, chosen on the device from a sample.  Every branch -- two passes, one
pass, list overflow behind a misleading sample -- must return what the plain-C oracle returns, bit for bit, and what
the library returns with CC_AMD_TWO_PASS=0."""
import os
import subprocess
import sys

import numpy as np
import pytest

from checkers import BCH, O1, O2, Oracle, awgn_llr

import channelcoding_amd as cc

pytestmark = pytest.mark.gpu

FRAMES = (1 << 18) + 777  # the launcher takes the two-pass route from 1.5e9 frame-edges on (2^18 frames of this code)


def decode_on_device(code, y):
    """Device-pointer entry point with the whole batch in one call (host buffers are cut into 32 MiB chunks, which
    never reach the two-pass route)."""
    import torch
    r = code.correct_batch(torch.from_numpy(y).cuda())
    return {k: v.cpu().numpy() for k, v in r.items()}


def batch(kind, n, rate):
    rng = np.random.default_rng({"high": 1, "low": 2, "trap": 3, "ordered": 4}[kind])
    zeros = np.zeros((FRAMES, n), np.uint8)
    if kind == "high":  # ~5 % of the frames need a second iteration: two passes
        return awgn_llr(rng, zeros, rate, 8.0)
    if kind == "low":   # most frames need many: one pass
        return awgn_llr(rng, zeros, rate, 4.0)
    if kind == "ordered":  # an SNR sweep in one batch, clean half first: a leading sample would choose two passes
        y = awgn_llr(rng, zeros, rate, 3.0)
        y[:FRAMES // 2] = awgn_llr(rng, zeros[:FRAMES // 2], rate, 11.0)
        return y
    # "trap": clean frames exactly where the launcher samples (four runs of 1024 frames, launch_two_pass) in a noisy
    # batch -> two passes are chosen and the list overflows
    y = awgn_llr(rng, zeros, rate, 3.0)
    hop = (FRAMES // 4) & ~63
    for r in range(4):
        y[r * hop:r * hop + 1024] = awgn_llr(rng, zeros[:1024], rate, 11.0)
    return y


@pytest.mark.parametrize("kind,variant,rule", [(k, v, r) for k in ("high", "low", "trap") for v, r in (("ms", O2), ("nms", O1))] +
                         [("ordered", "nms", O1)])
def test_two_pass_matches_oracle(kind, variant, rule):
    o = Oracle(BCH, 8, 3)
    tag = cc.min_sum_tag(20) if variant == "ms" else cc.normalized_min_sum_tag(20, 0.8)
    code = cc.primitive_bch(8, cc.errors(3), tag, stop_rule=rule)
    y = batch(kind, o.n, o.l / o.n)
    res = decode_on_device(code, y)
    # the oracle on a sample that contains every kind of frame: the first 300, and 300 that ran longest on the device
    pick = np.unique(np.concatenate([np.arange(300), np.argsort(res["iters"])[-300:], np.flatnonzero(res["status"])[:100]]))
    ob, _, oit, ost = o.minsum(0 if variant == "ms" else 1, 20, y[pick], 0.8 if variant == "nms" else 1.0, 0.0, rule,
                              fast=True)
    assert np.array_equal(res["status"][pick] != 0, ost != 0)
    assert np.array_equal(res["out"][pick], ob)
    assert np.array_equal(res["iters"][pick], oit)
    if kind == "high":
        assert (res["iters"] == 0).mean() > 0.9 and (res["iters"] > 0).any()
    # size-independent property on all frames: every frame reported converged satisfies the stop rule
    ok = res["status"] == 0
    if rule == O1:
        assert not res["out"][ok].any()
    else:
        H = np.array(code.H(), np.uint8)
        assert not ((res["out"][ok].astype(np.int32) @ H.T.astype(np.int32)) & 1).any()


def test_two_pass_equals_one_pass(tmp_path):
    """Same inputs with CC_AMD_TWO_PASS=0 in a process of its own (the switch is read once)."""
    here = os.path.dirname(os.path.abspath(__file__))
    script = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from test_gpu_twopass import batch, decode_on_device\n"
        "import channelcoding_amd as cc\n"
        "code = cc.primitive_bch(8, cc.errors(3), cc.min_sum_tag(20))\n"
        "for kind in ('high', 'trap'):\n"
        "    r = decode_on_device(code, batch(kind, 255, 231 / 255))\n"
        "    np.savez(sys.argv[1] + kind, out=r['out'], iters=r['iters'], status=r['status'])\n"
        % (here, os.path.dirname(here)))
    for mode in ("1", "0"):
        out = subprocess.run([sys.executable, "-c", script, str(tmp_path / ("m" + mode + "_"))],
                             env=dict(os.environ, CC_AMD_TWO_PASS=mode), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    for kind in ("high", "trap"):
        a = np.load(tmp_path / ("m1_" + kind + ".npz"))
        b = np.load(tmp_path / ("m0_" + kind + ".npz"))
        for key in ("out", "iters", "status"):
            assert np.array_equal(a[key], b[key]), (kind, key)


def test_monte_carlo_counters_do_not_depend_on_the_route():
    """cc_mc_run_dev with its operating-point shortcuts -- the pre-check route (mc.hip: frames whose channel hard decision
    is a codeword are counted where the noise is made) from ~5.7 dB on, two-pass decoding below -- against
    CC_AMD_TWO_PASS=0, the plain route: same noise (keyed by the global frame index), so every counter -- frames, word /
    bit / channel errors, failures, undetected errors, iteration sum and histogram -- must agree.  All-zero and random
    codewords, three stop rules, a scaled variant."""
    here = os.path.dirname(os.path.abspath(__file__))
    script = (
        "import sys; sys.path.insert(0, %r)\n"
        "import channelcoding_amd as cc\n"
        "from channelcoding_amd.montecarlo import DeviceBackend\n"
        "cases = [(cc.min_sum_tag(20), 2, False, 8.0, 1 << 21), (cc.min_sum_tag(20), 2, True, 8.0, 1 << 20),\n"
        "         (cc.min_sum_tag(20), 2, True, 6.0, 1 << 18), (cc.min_sum_tag(20), 1, False, 7.0, 1 << 18),\n"
        "         (cc.min_sum_tag(20), 1, True, 7.0, 1 << 17), (cc.min_sum_tag(20), 0, True, 7.0, 1 << 18),\n"
        "         (cc.normalized_min_sum_tag(10, (8, 10)), 2, True, 6.5, 1 << 18),\n"
        "         (cc.self_correcting_2_min_sum_tag(10), 2, True, 7.0, 1 << 18), (cc.min_sum_tag(20), 2, False, 4.0, 1 << 17)]\n"
        "for tag, rule, rnd, ebno, frames in cases:\n"
        "    be = DeviceBackend(cc.primitive_bch(8, cc.errors(3), tag, stop_rule=rule), random_codewords=rnd)\n"
        "    print('COUNTERS', [int(x) for x in be.run(ebno, 1234, 77, frames)])\n" % os.path.dirname(here))
    seen = []
    for mode in ("1", "0"):
        out = subprocess.run([sys.executable, "-c", script], env=dict(os.environ, CC_AMD_TWO_PASS=mode), capture_output=True,
                             text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
        seen.append([ln for ln in out.stdout.splitlines() if ln.startswith("COUNTERS")])
    assert len(seen[0]) == 9 and seen[0] == seen[1], [(a, b) for a, b in zip(*seen) if a != b][:2]
    counters = eval(seen[0][0].split(" ", 1)[1])
    assert counters[0] == 1 << 21 and 0 < counters[1] < counters[0] // 100  # frames; a few word errors at 8 dB
