"""GPU parity tests of the bit-plane path of GF(2^8) codes (csrc/bitslice.hip + the split chunk kernels): syndromes
as XOR networks on 32 frames per register, Berlekamp-Massey over 64-frame chunks, the correction kernel, and the
encoder by evaluation + interpolation.  Against the plain-C oracle, bit-exact, at batch sizes around every layout
boundary (32 frames per group, 64 per chunk, 2048 per block)."""
import numpy as np
import pytest

from checkers import BCH, BM, EUKLID, PGZ, RS, Oracle
from test_gpu_algebraic import TAGS, check_against_oracle, corrupt

import channelcoding_amd as cc

pytestmark = pytest.mark.gpu

SIZES = (1, 31, 32, 33, 63, 64, 65, 255, 256, 257, 700, 2047, 2048, 2049, 4161)  # (256 frames: a tile of the fused kernel)


def make(fam, t, alg=BM):
    return (cc.primitive_bch if fam == BCH else cc.rs)(8, cc.errors(t), TAGS[alg]())


@pytest.mark.parametrize("fam,t", [(RS, 16), (RS, 8), (RS, 4), (RS, 5), (RS, 1), (RS, 2), (RS, 3), (BCH, 4), (BCH, 9), (BCH, 1),
                                   (BCH, 2), (BCH, 3)])
def test_decode_at_layout_boundaries(fam, t):
    o = Oracle(fam, 8, t)
    code = make(fam, t)
    assert code.kernel_info()["kernel"].startswith("algebraic_chunk_kernel")
    rng = np.random.default_rng(1000 * fam + t)
    hi = 2 if fam == BCH else 256
    for frames in SIZES:
        if (frames > 700 or frames in (255, 256, 257)) and t not in (16, 4):
            continue
        cw = o.encode(rng.integers(0, hi, (frames, o.l)).astype(np.uint8))
        rx = np.stack([corrupt(rng, o, cw[f], int(rng.integers(0, o.t + 3))) for f in range(frames)])
        rx[frames // 2] = cw[frames // 2]  # a clean frame among dirty ones
        check_against_oracle(code.correct_batch(rx), o, BM, rx)


def test_pgz_tag_and_signed_input():
    """The PGZ tag (bounded-distance) and float input (bit = x < 0, cyclic.h:163-173) through the same kernels."""
    o = Oracle(BCH, 8, 6)
    rng = np.random.default_rng(77)
    frames = 2100
    cw = o.encode(rng.integers(0, 2, (frames, o.l)).astype(np.uint8))
    rx = np.stack([corrupt(rng, o, cw[f], int(rng.integers(0, o.t + 3))) for f in range(frames)])
    soft = np.where(rx != 0, -1.0, 1.0).astype(np.float32) * rng.uniform(0.1, 3.0, rx.shape).astype(np.float32)
    for alg in (BM, PGZ):
        code = make(BCH, 6, alg)
        a = code.correct_batch(rx)
        b = code.correct_batch(soft)
        for key in ("out", "status", "nerr"):
            assert np.array_equal(a[key], b[key]), key
        check_against_oracle(a, o, alg, rx)


def test_all_clean_and_all_dirty_chunks():
    o = Oracle(RS, 8, 16)
    code = make(RS, 16)
    rng = np.random.default_rng(5)
    cw = o.encode(rng.integers(0, 256, (256, o.l)).astype(np.uint8))
    res = code.correct_batch(cw)
    assert (res["status"] == 0).all() and (res["nerr"] == 0).all() and np.array_equal(res["out"], cw)
    rx = np.stack([corrupt(rng, o, cw[f], 16) for f in range(256)])
    res = code.correct_batch(rx)
    assert (res["status"] == 0).all() and (res["nerr"] == 16).all() and np.array_equal(res["out"], cw)
    rx = np.stack([corrupt(rng, o, cw[f], 40) for f in range(256)])  # far beyond the capability
    check_against_oracle(code.correct_batch(rx), o, BM, rx)


@pytest.mark.parametrize("t", [16, 8])
def test_encode_by_interpolation(t):
    """RS(255,223) / RS(255,239): the remainder of the division, computed as the interpolation of the evaluations at
    the 2t roots, equals the oracle's long division; extraction returns the message."""
    o = Oracle(RS, 8, t)
    code = make(RS, t)
    rng = np.random.default_rng(300 + t)
    for frames in SIZES:
        msg = rng.integers(0, 256, (frames, o.l)).astype(np.uint8)
        msg[0] = 0
        if frames > 2:
            msg[1] = 255
            msg[2, :-1] = 0  # a single non-zero symbol at the top position
        cw = code.encode_batch(msg)
        assert np.array_equal(cw, o.encode(msg)), frames
        assert np.array_equal(code.extract_batch(cw), msg)


def test_table_kernels_stay_exact():
    """CC_AMD_NO_BITSLICE=1 sends the same codes through the table kernels the bit-plane path replaced (the chunked
    kernel of DESIGN 4.3b); the switch is read once per process, so the comparison runs in one of its own."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    script = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from checkers import BCH, BM, PGZ, RS, Oracle\n"
        "from test_gpu_algebraic import TAGS, check_against_oracle, corrupt\n"
        "import channelcoding_amd as cc\n"
        "rng = np.random.default_rng(4)\n"
        "for fam, t, alg in ((RS, 16, BM), (RS, 8, PGZ), (BCH, 6, BM)):\n"
        "    o = Oracle(fam, 8, t)\n"
        "    code = (cc.primitive_bch if fam == BCH else cc.rs)(8, cc.errors(t), TAGS[alg]())\n"
        "    hi = 2 if fam == BCH else 256\n"
        "    for frames in (1, 65, 2200):\n"
        "        cw = o.encode(rng.integers(0, hi, (frames, o.l)).astype(np.uint8))\n"
        "        rx = np.stack([corrupt(rng, o, cw[f], int(rng.integers(0, t + 6))) for f in range(frames)])\n"
        "        check_against_oracle(code.correct_batch(rx), o, alg, rx)\n"
        "print('ALT OK')\n" % (here, os.path.dirname(here)))
    out = subprocess.run([sys.executable, "-c", script], env=dict(os.environ, CC_AMD_NO_BITSLICE="1"),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ALT OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_small_calls_at_default_settings():
    """Without CC_AMD_PLANES_MIN_WORK (the suite sets it to 0, conftest.py) small calls of the bit-plane codes run one
    wavefront per frame and large ones the chain: same results as the oracle on both sides of the switch."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    script = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from checkers import BCH, BM, RS, Oracle\n"
        "from test_gpu_algebraic import TAGS, check_against_oracle, corrupt\n"
        "import channelcoding_amd as cc\n"
        "rng = np.random.default_rng(14)\n"
        "for fam, t, sizes in ((RS, 16, (1, 65, 2200, 12287, 12288)), (BCH, 3, (33, 65535, 65536))):\n"
        "    o = Oracle(fam, 8, t)\n"
        "    code = (cc.primitive_bch if fam == BCH else cc.rs)(8, cc.errors(t), TAGS[BM]())\n"
        "    hi = 2 if fam == BCH else 256\n"
        "    for frames in sizes:\n"
        "        cw = o.encode(rng.integers(0, hi, (min(frames, 700), o.l)).astype(np.uint8))\n"
        "        rx = np.stack([corrupt(rng, o, cw[f], int(rng.integers(0, t + 3))) for f in range(cw.shape[0])])\n"
        "        rx = np.tile(rx, ((frames + 699) // 700, 1))[:frames]  # (the oracle sees 700 distinct frames)\n"
        "        res = code.correct_batch(rx)\n"
        "        head = {k: v[:rx[:700].shape[0]] for k, v in res.items()}\n"
        "        check_against_oracle(head, o, BM, rx[:700])\n"
        "        for k in ('out', 'status', 'nerr'):\n"
        "            rep = np.tile(head[k], ((frames + 699) // 700,) + (1,) * (head[k].ndim - 1))[:frames]\n"
        "            assert np.array_equal(res[k], rep), (fam, frames, k)\n"
        "print('DEFAULT OK')\n" % (here, os.path.dirname(here)))
    env = {k: v for k, v in os.environ.items() if k != "CC_AMD_PLANES_MIN_WORK"}
    out = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "DEFAULT OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.parametrize("fam,t,alg", [(RS, 16, BM), (RS, 16, EUKLID), (RS, 8, BM), (RS, 5, EUKLID), (BCH, 4, BM), (BCH, 9, EUKLID),
                                       (RS, 2, BM)])
def test_erasures_on_the_plane_chain(fam, t, alg):
    """Erasure decoding on the bit-plane chain (round 3: chunk_bm_kernel pre-loads the erasure locator per lane and
    starts the recurrence at i = rho; locators up to degree 16 are corrected lane-per-frame, longer ones and binary
    codes go through chunk_fix_kernel; the Euklid tag runs the same chain as bounded-distance BM and then Sugiyama's own
    kernel over the frames it left undecoded -- the reference decodes some of those when rho is odd): 0 .. 2t erasures per frame (beyond 2t: CC_FRAME_ERASURES, checked at the end), errors up to
    and beyond the capability, clean frames with erasures, ragged batch sizes; against the oracle frame by frame."""
    from test_gpu_algebraic import check_erasure_frames
    o = Oracle(fam, 8, t)
    code = make(fam, t, alg)
    assert code.kernel_info()["kernel"].startswith("algebraic_chunk_kernel")
    rng = np.random.default_rng(31 * t + alg)
    hi = 2 if fam == BCH else 256
    for frames in (1, 65, 333):
        cw = o.encode(rng.integers(0, hi, (frames, o.l)).astype(np.uint8))
        rx = cw.copy()
        ers = []
        for f in range(frames):
            ne = int(rng.integers(0, 2 * o.t + 1))
            er = sorted(rng.choice(o.n, ne, replace=False).tolist())
            for e in er:
                rx[f, e] = int(rng.integers(0, hi))  # (an erased symbol may be right by chance)
            free = np.setdiff1d(np.arange(o.n), er)
            room = max(0, (2 * o.t - ne) // 2)
            for p in rng.choice(free, int(rng.integers(0, room + 2)), replace=False):
                rx[f, p] ^= 1 if fam == BCH else int(rng.integers(1, hi))
            ers.append(er)
        ers[0] = []  # a frame without erasures among them
        if frames > 1:
            rx[1] = cw[1]  # a clean frame that carries erasures
        check_erasure_frames(code, o, alg, rx, ers)
    # more than 2t erasures: CC_FRAME_ERASURES and the word untouched (the device's own class for it, as on the
    # one-wavefront-per-frame kernel; the reference's BM path fails later with another text) -- unless the word is clean
    cw = o.encode(rng.integers(0, hi, (40, o.l)).astype(np.uint8))
    rx = cw.copy()
    rx[::2, 7] ^= 1
    ers = [sorted(rng.choice(o.n, 2 * o.t + 1 + (f % 3), replace=False).tolist()) for f in range(40)]
    res = code.correct_batch(rx, erasures=ers)
    assert (res["status"][::2] == 4).all() and (res["nerr"][::2] == -1).all() and np.array_equal(res["out"], rx)
    assert (res["status"][1::2] == 0).all() and (res["nerr"][1::2] == 0).all()


def test_erasure_paths_agree():
    """profiles/tools/erasure_soak.py: 2^14 seeded frames x 8 configurations (RS / BCH, BM / Euklid tag; 0 .. 2t + 2
    erasures, errors up to and beyond the capability, clean frames with erasures) through the bit-plane chain and through
    the one-wavefront-per-frame kernel, in a process each (the switch is read once): the same words, counts and status
    of every frame (SHA-256 of the three arrays) -- all four failure classes occur."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "profiles", "tools", "erasure_soak.py")
    outs = []
    for work in ("0", "1000000000000"):
        r = subprocess.run([sys.executable, tool, "14"], env=dict(os.environ, CC_AMD_PLANES_MIN_WORK=work),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if "digest" in l]
        assert len(lines) == 8 and all(" rc 0 " in l for l in lines), r.stdout[-2000:]
        outs.append(lines)
    assert outs[0] == outs[1], "\n".join(a + "\n" + b for a, b in zip(*outs) if a != b)
    assert any("erasures 0 " not in l for l in outs[0]) and any("recheck 0 " not in l for l in outs[0])


@pytest.mark.parametrize("alg", [BM, EUKLID])
def test_erasure_calls_without_count_and_status_buffers(alg):
    """nerr / status may be NULL (channelcoding_amd.h): the erasure chain -- and the Euklid tag's second stage, which
    reads the status the chain left -- must return the same words without them."""
    import ctypes as C

    import torch
    from channelcoding_amd import capi
    o = Oracle(RS, 8, 16)
    code = make(RS, 16, alg)
    rng = np.random.default_rng(900 + alg)
    frames = 20000
    base = o.encode(rng.integers(0, 256, (250, o.l)).astype(np.uint8))
    rx, ers = base.copy(), []
    for f in range(250):
        ne = int(rng.integers(0, 34))
        er = sorted(rng.choice(o.n, ne, replace=False).tolist())
        for e in er:
            rx[f, e] = 0
        free = np.setdiff1d(np.arange(o.n), er)
        for p in rng.choice(free, int(rng.integers(0, max(1, (32 - ne) // 2 + 2))), replace=False):
            rx[f, p] ^= int(rng.integers(1, 256))
        ers.append(er)
    reps = frames // 250
    rxb = np.tile(rx, (reps, 1))
    pos = np.array([e for _ in range(reps) for er in ers for e in er], np.uint16)
    cnt = np.array([len(er) for _ in range(reps) for er in ers], np.int64)
    off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint32)
    lib = capi.lib()
    vp = lambda x: C.c_void_p(x.data_ptr())
    d_in = torch.from_numpy(rxb).cuda()
    d_pos = torch.from_numpy(pos.view(np.int16)).cuda()
    d_off = torch.from_numpy(off.view(np.int32)).cuda()
    outs = [torch.empty_like(d_in) for _ in range(2)]
    ne_ = torch.empty(frames, dtype=torch.int32, device="cuda")
    st_ = torch.empty(frames, dtype=torch.int32, device="cuda")
    sh = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.cc_correct_hard_batch_dev(code._h, vp(d_in), vp(d_pos), vp(d_off), vp(outs[0]), vp(ne_), vp(st_), frames, sh) == 0
    assert lib.cc_correct_hard_batch_dev(code._h, vp(d_in), vp(d_pos), vp(d_off), vp(outs[1]), None, None, frames, sh) == 0
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    head = dict(out=outs[0][:250].cpu().numpy(), nerr=ne_[:250].cpu().numpy(), status=st_[:250].cpu().numpy())
    for f in range(250):
        out, nerr, st, ub = o.correct_hard(alg, rx[f], ers[f])
        assert (head["status"][f] == 0) == (st[0] == 0), (alg, f, len(ers[f]))
        if st[0] == 0:
            assert np.array_equal(head["out"][f], out[0]) and head["nerr"][f] == nerr[0], (alg, f)
