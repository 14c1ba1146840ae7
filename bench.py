#!/usr/bin/env python3
"""bench.py -- headline benchmark: BCH(255,231,t=3) min-sum decode, batch 2^20 frames per GPU.

One "step" = one pass of the hot path (cc_correct_soft_batch_dev: LLR frames resident in HBM ->
hard decisions + iteration index + status) over one batch.  Prints ONE JSON line on rank 0.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--ebno 4.0] [--batch-log2 20]
  (N > 1: launched by torch.distributed.run, one rank per GPU; frames are independent, so ranks
   shard the batch with no data-path collective -> weak scaling; timing = max over ranks.)

roofline.achieved = algorithmic HBM bytes per launch (4n LLR bytes in + n hard bytes out + 6 B
iters/status per frame) / average kernel duration measured with HIP events on the
launch stream.  cpu_baseline = the real reference (oracle/_ref, built from /root/reference in the
dev container) or, if that prebuilt library is absent, the plain-C oracle, timed on a bounded sample
of the same frames on one host core.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); ~6300 measured attainable

# The bound that actually holds: VALU issue.  The instruction mix of the kernel that runs is
# read from profiles/kernel_valu_mix.json, keyed by the name cc_kernel_info reports: a summary committed together
# with the ISA count / PMC pass it came from (profiles/tools/isa_mix.py), never constants in this file.  A kernel
# without an entry gets no valu_issue object.
SIMDS, CLOCK_HZ = 256 * 4, 2.4e9


def valu_mix(kernel):
    try:
        with open(os.path.join(ROOT, "profiles", "kernel_valu_mix.json")) as f:
            return json.load(f).get(kernel)
    except (OSError, ValueError):
        return None


def measured_traffic(kernel, batch_log2):
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE x2 on gfx950 +
    WRITE_SIZE; separate passes).  None when no measurement exists for this kernel and batch size."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic_minsum.json")) as f:
            t = json.load(f)
        if t["kernel"] == kernel and t["batch_log2"] == batch_log2:
            return t["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def rs_traffic():
    """HBM bytes of one RS(255,223) decode call of 2^20 frames from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE over
    the call's kernels, profiles/traffic_rs.json <- profiles/tools/r03_collect.py); None without a measurement."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic_rs.json")) as f:
            return json.load(f)["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def secondary_workloads(torch, cc, capi, dev):
    """The other single-GPU BASELINE configs, reported next to the headline (not part of `value`):
    configs[1] BCH(63,45) MS<10> at 4 dB, batch 2^16; configs[3] RS(255,223) syndrome + BM + root search +
    error values, batch 2^20, e ~ U{0..16} symbol errors."""
    lib = capi.lib()
    out = {}
    vp = lambda t: C.c_void_p(t.data_ptr())
    sh = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    g = torch.Generator(device=dev)
    g.manual_seed(99)
    # configs[1]
    code = cc.primitive_bch(6, cc.errors(3), cc.min_sum_tag(10))
    B = 1 << 16
    y = torch.empty((B, 63), dtype=torch.float32, device=dev).normal_(1.0, float(code.sigma(4.0)), generator=g)
    hard = torch.empty((B, 63), dtype=torch.uint8, device=dev)
    it = torch.empty(B, dtype=torch.int16, device=dev)
    st = torch.empty(B, dtype=torch.int32, device=dev)
    ms = timed(lambda: lib.cc_correct_soft_batch_dev(code._h, vp(y), None, None, vp(hard), None, vp(it), vp(st), B, sh))
    out["bch63_45_ms10_4dB_2^16"] = {"frames_per_s": B / (ms * 1e-3), "kernel_ms": ms,
                                     "converged_fraction": float((st == 0).float().mean())}
    B = 1 << 20  # the same code at the headline batch size (2^16 frames are a 0.16 ms launch)
    y = torch.empty((B, 63), dtype=torch.float32, device=dev).normal_(1.0, float(code.sigma(4.0)), generator=g)
    hard = torch.empty((B, 63), dtype=torch.uint8, device=dev)
    it = torch.empty(B, dtype=torch.int16, device=dev)
    st = torch.empty(B, dtype=torch.int32, device=dev)
    ms = timed(lambda: lib.cc_correct_soft_batch_dev(code._h, vp(y), None, None, vp(hard), None, vp(it), vp(st), B, sh))
    out["bch63_45_ms10_4dB_2^20"] = {"frames_per_s": B / (ms * 1e-3), "kernel_ms": ms,
                                     "achieved_GBs": 321 * B / (ms * 1e-3) / 1e9}
    # headline code at other operating points (4 dB and 6 dB, a random-codeword variant under
    # O2, and the as-shipped stop rule O0 = exactly one iteration per frame)
    B = 1 << 20
    y = torch.empty((B, 255), dtype=torch.float32, device=dev)
    hard = torch.empty((B, 255), dtype=torch.uint8, device=dev)
    it = torch.empty(B, dtype=torch.int16, device=dev)
    st = torch.empty(B, dtype=torch.int32, device=dev)

    def headline_point(key, ebno, stop_rule=2, random_codewords=False, tag=None):
        code = cc.primitive_bch(8, cc.errors(3), tag or cc.min_sum_tag(20), stop_rule=stop_rule)
        y.normal_(0.0, float(code.sigma(ebno)), generator=g)
        if random_codewords:  # y = (1 - 2c) + sigma N
            msg = torch.randint(0, 2, (B, code.l), dtype=torch.uint8, device=dev, generator=g)
            cw = code.encode_batch(msg)
            y.add_(1.0 - 2.0 * cw.to(torch.float32))
        else:
            y.add_(1.0)
        ms = timed(lambda: lib.cc_correct_soft_batch_dev(code._h, vp(y), None, None, vp(hard), None, vp(it), vp(st),
                                                         B, sh))
        run = torch.where(st == 0, it.to(torch.int32) + 1, it.to(torch.int32))
        res = {"frames_per_s": B / (ms * 1e-3), "kernel_ms": ms, "mean_iterations_run": float(run.float().mean()),
               "converged_fraction": float((st == 0).float().mean()), "achieved_GBs": 1281 * B / (ms * 1e-3) / 1e9}
        if random_codewords:
            res["decoded_equals_sent_fraction"] = float((hard == cw).all(dim=1).float().mean())
        out[key] = res

    headline_point("bch255_231_ms20_6dB_2^20", 6.0)
    headline_point("bch255_231_ms20_8dB_2^20", 8.0)
    headline_point("bch255_231_ms20_4dB_random_codewords_2^20", 4.0, random_codewords=True)
    headline_point("bch255_231_ms20_4dB_stop_rule_O0_as_shipped_2^20", 4.0, stop_rule=0)
    headline_point("bch255_231_nms20_0.8_4dB_2^20", 4.0, tag=cc.normalized_min_sum_tag(20, 0.8))
    headline_point("bch255_231_scms1_20_4dB_2^20", 4.0, tag=cc.self_correcting_1_min_sum_tag(20))
    headline_point("bch255_231_scms2_20_4dB_2^20", 4.0, tag=cc.self_correcting_2_min_sum_tag(20))
    del y, hard
    # configs[3]
    rs = cc.rs(8, cc.errors(16), cc.berlekamp_massey_tag())
    B = 1 << 20
    msg = torch.randint(0, 256, (B, rs.l), dtype=torch.uint8, device=dev, generator=g)
    cw = rs.encode_batch(msg)
    nerr = torch.randint(0, 17, (B,), device=dev, generator=g)
    perm = torch.rand((B, 255), device=dev, generator=g).argsort(dim=1)[:, :16]
    vals = torch.randint(1, 256, (B, 16), dtype=torch.uint8, device=dev, generator=g)
    vals = torch.where(torch.arange(16, device=dev)[None, :] < nerr[:, None], vals, torch.zeros_like(vals))
    rx = cw.clone()
    rx.scatter_(1, perm, rx.gather(1, perm) ^ vals)
    del perm, vals, msg
    outw = torch.empty_like(rx)
    ne = torch.empty(B, dtype=torch.int32, device=dev)
    st = torch.empty(B, dtype=torch.int32, device=dev)
    ms = timed(lambda: lib.cc_correct_hard_batch_dev(rs._h, vp(rx), None, None, vp(outw), vp(ne), vp(st), B, sh))
    ok = bool(torch.equal(outw, cw)) and int((st != 0).sum()) == 0
    out["rs255_223_bm_2^20"] = {"frames_per_s": B / (ms * 1e-3), "kernel_ms": ms, "all_frames_corrected": ok,
                                "achieved_GBs": 518 * B / (ms * 1e-3) / 1e9, "algorithmic_bytes_per_frame": 518,
                                "traffic": rs_traffic()}
    rse = cc.rs(8, cc.errors(16), cc.euklid_tag())  # same frames under the Euklid tag (bounded-distance BM on the planes)
    ms_e = timed(lambda: lib.cc_correct_hard_batch_dev(rse._h, vp(rx), None, None, vp(outw), vp(ne), vp(st), B, sh))
    out["rs255_223_euklid_2^20"] = {"frames_per_s": B / (ms_e * 1e-3), "kernel_ms": ms_e,
                                    "all_frames_corrected": bool(torch.equal(outw, cw)) and int((st != 0).sum()) == 0}
    msg_again = cw[:, rs.k:].contiguous()  # (the message, resident in HBM like every other input of the timed calls)
    enc_ms = timed(lambda: rs.encode_batch(msg_again))
    out["rs255_223_encode_2^20"] = {"frames_per_s": B / (enc_ms * 1e-3), "kernel_ms": enc_ms}
    # errors and erasures (BM tag): 4 erased positions (zeroed, passed as CSR) + 0 .. 6 errors elsewhere per frame
    rho, maxe = 4, 6
    pos = torch.rand((B, rs.n), device=dev, generator=g).argsort(dim=1)[:, :rho + maxe]
    nerr = torch.randint(0, maxe + 1, (B,), device=dev, generator=g)
    vals = torch.randint(1, 256, (B, maxe), dtype=torch.uint8, device=dev, generator=g)
    vals = torch.where(torch.arange(maxe, device=dev)[None, :] < nerr[:, None], vals, torch.zeros_like(vals))
    rxe = cw.clone()
    rxe.scatter_(1, pos[:, rho:], rxe.gather(1, pos[:, rho:]) ^ vals)
    rxe.scatter_(1, pos[:, :rho], torch.zeros((B, rho), dtype=torch.uint8, device=dev))
    er = pos[:, :rho].sort(dim=1).values.to(torch.int16).contiguous().view(-1)
    off = (torch.arange(B + 1, device=dev, dtype=torch.int64) * rho).to(torch.int32)
    ms_x = timed(lambda: lib.cc_correct_hard_batch_dev(rs._h, vp(rxe), vp(er), vp(off), vp(outw), vp(ne), vp(st), B, sh))
    out["rs255_223_bm_4_erasures_2^20"] = {"frames_per_s": B / (ms_x * 1e-3), "kernel_ms": ms_x,
                                           "all_frames_corrected": bool(torch.equal(outw, cw)) and int((st != 0).sum()) == 0}
    return out


def verify_batch(torch, code, y, hard, iters, status, iterations, stop_rule, sample=512):
    """Correctness gate of the headline number, OUTSIDE the timed region (VERDICT r1 Weak #3).

    (i)  every frame the decoder reports as converged under the GF(2) stop rule carries a codeword: H b^T = 0 over
         GF(2) for all of them (all B frames, on the device, H from cc_get_H);
    (ii) a strided sample of `sample` frames is decoded by the CPU checker (oracle/cc_oracle.c, orc_minsum_fast --
         test infrastructure, imported here as the checker only) and must agree bit for bit in hard decisions,
         iteration index and status."""
    from checkers import BCH, Oracle
    B, n = y.shape
    res = {"verified": False, "frames_checked_parity": 0, "parity_violations": None,
           "oracle_sample": sample, "oracle_mismatches": None}
    H = torch.from_numpy(code.H().astype(np.float32)).to(y.device)  # k x n
    if stop_rule == 2:
        bad = 0
        for lo in range(0, B, 1 << 18):
            hb = hard[lo:lo + (1 << 18)].to(torch.float32) @ H.t()       # exact: row weight < 2^24
            odd = (hb.to(torch.int32) & 1).any(dim=1)
            bad += int((odd & (status[lo:lo + (1 << 18)] == 0)).sum())
        res["frames_checked_parity"] = int((status == 0).sum())
        res["parity_violations"] = bad
    idx = torch.arange(0, B, max(1, B // sample), device=y.device)[:sample]
    ys = y[idx].cpu().numpy()
    ob, _, oit, ost = Oracle(BCH, code.q, code.t).minsum(0, iterations, ys, stop=stop_rule, fast=True)
    mism = int((hard[idx].cpu().numpy() != ob).any(axis=1).sum())
    mism += int((status[idx].cpu().numpy() != ost).sum())
    conv = ost == 0
    mism += int((iters[idx].cpu().numpy().astype(np.int64)[conv] != oit.astype(np.int64)[conv]).sum())
    res["oracle_mismatches"] = mism
    res["verified"] = (mism == 0) and (res["parity_violations"] in (0, None))
    return res


CPU_WORKER = r"""
import sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
from checkers import BCH, O2, Oracle, RefLib
y = np.load(sys.argv[2], mmap_mode="r")
k, m, iterations = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
part = np.ascontiguousarray(y[k * m:(k + 1) * m])
t0 = time.perf_counter()
if RefLib.available():
    RefLib.get(1).minsum(6, 0, iterations, 1, part)
else:
    Oracle(BCH, 8, 3).minsum(0, iterations, part, stop=O2, fast=True)
print(time.perf_counter() - t0)
"""


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(y_sample, iterations, budget_s=12.0):
    """Reference CPU path on this box's host cores, same frames as the GPU workload, stop rule O2.  The reference
    decodes one frame at a time on one thread (src/simulation/simulation.c++:124-136) and gets its parallelism from
    running one simulation per pool thread (benchmark.c++:435-439).  The baseline gives it one PROCESS per core
    (the reference allocates heavily per frame; threads of one process serialise in malloc), each decoding its own
    slice of the sample through the reference library; value = frames / slowest worker."""
    import subprocess
    import tempfile
    from checkers import BCH, O2, Oracle, RefLib
    cores = max(1, min(16, os.cpu_count() or 1))  # a one-GPU box gives its job a 16-CPU share
    if RefLib.available():
        ref = RefLib.get(1)
        t0 = time.perf_counter()
        ref.minsum(6, 0, iterations, 1, y_sample[:8])  # calibrate, then size the per-core slice to the budget
        per = max((time.perf_counter() - t0) / 8, 1e-4)
        m = int(max(16, min(len(y_sample) // cores, budget_s / per)))
        kind = "reference"
        what = ("stop rule O2: oracle/_ref/libccref_o1.so is the reference built with matrix::end() repaired "
                "(matrix.h:50); utype = 1 instantiates min_sum<float, ef_element<2,1>>(H(), y, min_sum_tag<%d>), "
                "i.e. the GF(2) parity stop test; H rebuilt per frame" % iterations)
    else:
        m = min(len(y_sample) // cores, 2000)
        kind, what = "port", "oracle/cc_oracle.c orc_minsum_fast (O(w) restatement)"
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "sample.npy")
        np.save(path, y_sample[:cores * m])
        procs = [subprocess.Popen([sys.executable, "-c", CPU_WORKER, os.path.join(ROOT, "tests"), path, str(k), str(m),
                                   str(iterations)], stdout=subprocess.PIPE, text=True) for k in range(cores)]
        secs = [float(p.communicate()[0].strip().splitlines()[-1]) for p in procs]
    sec = max(secs)
    return dict(value=cores * m / sec, unit="frames/s", cores=cores, kind=kind, per_core=m / (sum(secs) / cores),
                stop_rule="O2",
                sample="%d processes x %d of the benchmark's frames, slowest %.1f s, %s" % (cores, m, sec, what),
                host_cpus=os.cpu_count(), cpu_model=cpu_model())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--ebno", type=float, default=4.0)
    ap.add_argument("--batch-log2", type=int, default=20)
    ap.add_argument("--iterations", type=int, default=20)
    ap.add_argument("--stop-rule", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    # rehearsal of the N > 1 flow on a one-GPU box: control collectives over gloo, every rank on device 0
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--single-device", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: start the ranks ourselves, as a CHILD process and before this process
        # has imported torch or touched the GPU (never exec once a GPU is initialised), and leave with its code
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29541"),
               os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.pop("MASTER_PORT", None)
        sys.exit(subprocess.run(cmd, env=env).returncode)

    import torch
    import channelcoding_amd as cc
    from channelcoding_amd import capi

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the decoder has no CPU path")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))

    dev = torch.device("cuda", local_rank)
    B = 1 << args.batch_log2
    code = cc.primitive_bch(8, cc.errors(3), cc.min_sum_tag(args.iterations), stop_rule=args.stop_rule)
    n = code.n
    sigma = code.sigma(args.ebno)

    # synthetic input: all-zero codeword (as the reference's awgn_simulation, simulation.c++:113-125),
    # y = 1 + sigma*N(0,1) in f32; each rank owns a different shard of the global frame sequence.
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    y = torch.empty((B, n), dtype=torch.float32, device=dev)
    y.normal_(mean=1.0, std=float(sigma), generator=gen)
    hard = torch.empty((B, n), dtype=torch.uint8, device=dev)
    iters = torch.empty(B, dtype=torch.int16, device=dev)
    status = torch.empty(B, dtype=torch.int32, device=dev)
    lib = capi.lib()
    stream = torch.cuda.current_stream(dev)
    sh = C.c_void_p(stream.cuda_stream)
    vp = lambda t: C.c_void_p(t.data_ptr())

    def step():
        rc = lib.cc_correct_soft_batch_dev(code._h, vp(y), None, None, vp(hard), None, vp(iters), vp(status), B, sh)
        if rc != 0:
            raise cc.CcError(rc, "cc_correct_soft_batch_dev")

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream)
        step()
        b.record(stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    # attainable HBM bandwidth on this box: device-to-device copy of the LLR batch (read + write bytes)
    scratch = torch.empty_like(y)
    scratch.copy_(y)
    torch.cuda.synchronize()
    ca, cb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ca.record(stream)
    for _ in range(5):
        scratch.copy_(y)
    cb.record(stream)
    torch.cuda.synchronize()
    copy_gbs = 2 * y.numel() * 4 * 5 / (ca.elapsed_time(cb) * 1e-3) / 1e9
    del scratch

    verification = verify_batch(torch, code, y, hard, iters, status, args.iterations, args.stop_rule)
    if dist is not None:  # every rank's shard must pass
        okt = torch.tensor([1 if verification["verified"] else 0], dtype=torch.int64,
                           device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        verification["verified"] = bool(okt.item())

    it_host = iters.to(torch.int32)
    conv = int((status == 0).sum().item())
    iters_run = torch.where(status == 0, it_host + 1, it_host)  # iterations actually executed
    mean_iters = float(iters_run.float().mean().item())
    histogram = torch.bincount(iters_run.to(torch.int64), minlength=args.iterations + 1).tolist()

    if rank == 0:
        frames_per_s = world * B * args.steps / elapsed
        bytes_per_frame = 5 * n + 6
        achieved = bytes_per_frame * B / (kernel_ms * 1e-3) / 1e9
        name = C.create_string_buffer(128)
        fpw, thr, ldsb = C.c_uint32(), C.c_uint32(), C.c_uint32()
        lib.cc_kernel_info(code._h, name, 128, C.byref(fpw), C.byref(thr), C.byref(ldsb))
        out = {
            "metric": "decoded frames/sec (coded bits/sec = value x 255), BCH(255,231) min-sum, batch=2^%d" % args.batch_log2,
            "value": frames_per_s if verification["verified"] else None,  # a number for wrong output is no number
            "unit": "frames/s",
            "coded_bits_per_sec": frames_per_s * n if verification["verified"] else None,
            "verified": verification["verified"],
            "verification": verification,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BCH(255,231,t=3) min-sum MS<%d>, stop rule O%d, all-zero codeword + AWGN Eb/N0=%.1f dB "
                            "(sigma=%.6f), batch=2^%d frames per GPU, LLR f32 resident in HBM -> hard bytes + iters + status"
                            % (args.iterations, args.stop_rule, args.ebno, sigma, args.batch_log2),
                "frames_per_gpu": B, "n": n, "iterations_max": args.iterations,
                "mean_iterations_run": mean_iters, "converged_fraction": conv / B,
                "iterations_run_histogram": histogram,  # index = iterations executed by a frame (rank 0's shard)
                "parallelism": "frames sharded over %d GPU(s), no data-path collective" % world,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(name.value.decode(), args.batch_log2),
                "kernel": name.value.decode(), "kernel_ms": kernel_ms, "algorithmic_bytes_per_frame": bytes_per_frame,
                "measured_copy_GBs": copy_gbs,  # attainable HBM rate on this box (d2d copy, read + write)
                "note": "the path is bound by instruction issue (one instruction per ~7 cycles and wavefront at two waves "
                        "per SIMD, profiles/r03_experiments.md E14), not by HBM (SURVEY F4): %.1f min-sum iterations per "
                        "frame on average" % mean_iters,
            },
        }
        mix = valu_mix(name.value.decode())
        if mix is not None:
            cyc = mix["simd_issue_cycles_per_frame_iteration"]
            frame_iters = B * mean_iters / (kernel_ms * 1e-3)
            out["roofline"]["valu_issue"] = {
                "simd_cycles_per_frame_iteration": cyc, "frame_iterations_per_s": frame_iters,
                "achieved_simd_cycles_per_s": frame_iters * cyc, "peak_simd_cycles_per_s": SIMDS * CLOCK_HZ,
                "frac": frame_iters * cyc / (SIMDS * CLOCK_HZ),
                "mix": mix,
                "note": "fraction of all SIMD issue cycles spent issuing this kernel's VALU instructions; mix from "
                        "profiles/kernel_valu_mix.json",
            }
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only: the other ranks would idle at the barrier
            out["cpu_baseline"] = cpu_baseline(y[:32768].cpu().numpy(), args.iterations)
        else:
            out["cpu_baseline"] = None
        if world == 1 and not args.no_secondary:
            del y, hard
            out["secondary"] = secondary_workloads(torch, cc, capi, dev)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
