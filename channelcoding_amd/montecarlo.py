"""Batched AWGN Monte-Carlo harness, sharded over the GPUs of one node.

Replaces ``awgn_simulation`` (src/simulation/simulation.h:71-83, simulation.c++:83-150 of the
reference).  What is kept: the Eb/N0 ladder (start just above the Shannon limit of the code's rate,
``step`` dB apart, up to max(8, start)), the adaptive sample count ``min(1e6, 5e3 / wer)`` seeded with
wer = 0.5 (simulation.c++:91-93,:110), sigma = 1/sqrt(2 R 10^(EbN0/10)) (:83-85), word-error counting
(:128-135) and the two-column log file "<to_string()>.log" with the reference's field widths (:96-103,
:145-148; an existing file is refused, :72-81).

What is new: frames of one point are independent, so rank r of W decodes the contiguous range
[r*F/W, (r+1)*F/W) of the *global* frame index on its own GPU (cc_mc_run_dev generates the noise on
device from (seed, global frame index)), and ONE all-reduce (RCCL over xGMI for CUDA tensors, gloo on
CPU) of the 64-word counter vector per point gives every rank the totals -- which the adaptive sample
count of the next point needs.  Totals are bit-identical for any number of ranks.

The ladder's start point follows the reference's own Shannon-limit look-up ``ebno()`` (simulation.c++:21-70)
including its indexing (see `reference_ebno`), so every "<decoder>.log" starts on the line the reference's does.
"""
import ctypes as C
import math
import os

import numpy as np

from . import _capi as capi

COUNTER_NAMES = {
    "frames": capi.MC_FRAMES, "word_errors": capi.MC_WORD_ERRORS, "bit_errors": capi.MC_BIT_ERRORS,
    "failures": capi.MC_FAILURES, "undetected": capi.MC_UNDETECTED, "iter_sum": capi.MC_ITER_SUM,
    "channel_bit_errors": capi.MC_CHANNEL_BIT_ERRORS,
}


# Shannon limit of the BPSK-AWGN channel (dB) for the code rates of RATES -- the data of simulation.c++:21-52:
# rates 0.01 .. 0.80 in steps of 0.01, then 51 unevenly spaced rates up to 0.999.
RATES = tuple(i / 100.0 for i in range(1, 81)) + (
    0.807, 0.817, 0.827, 0.837, 0.846, 0.855, 0.864, 0.872, 0.880, 0.887, 0.894, 0.900, 0.907, 0.913, 0.918, 0.924,
    0.929, 0.934, 0.938, 0.943, 0.947, 0.951, 0.954, 0.958, 0.961, 0.964, 0.967, 0.970, 0.972, 0.974, 0.976, 0.978,
    0.980, 0.982, 0.983, 0.984, 0.985, 0.986, 0.987, 0.988, 0.989, 0.990, 0.991, 0.992, 0.993, 0.994, 0.995, 0.996,
    0.997, 0.998, 0.999)
LIMITS = (
    -1.548, -1.531, -1.500, -1.470, -1.440, -1.409, -1.378, -1.347, -1.316, -1.285, -1.254, -1.222, -1.190, -1.158,
    -1.126, -1.094, -1.061, -1.028, -0.995, -0.963, -0.928, -0.896, -0.861, -0.827, -0.793, -0.757, -0.724, -0.687,
    -0.651, -0.616, -0.579, -0.544, -0.507, -0.469, -0.432, -0.394, -0.355, -0.314, -0.276, -0.236, -0.198, -0.156,
    -0.118, -0.074, -0.032, 0.010, 0.055, 0.097, 0.144, 0.188, 0.233, 0.279, 0.326, 0.374, 0.424, 0.474, 0.526, 0.574,
    0.628, 0.682, 0.734, 0.791, 0.844, 0.904, 0.960, 1.021, 1.084, 1.143, 1.208, 1.275, 1.343, 1.412, 1.483, 1.554,
    1.628, 1.708, 1.784, 1.867, 1.952, 2.045, 2.108, 2.204, 2.302, 2.402, 2.503, 2.600, 2.712, 2.812, 2.913, 3.009,
    3.114, 3.205, 3.312, 3.414, 3.500, 3.612, 3.709, 3.815, 3.906, 4.014, 4.115, 4.218, 4.304, 4.425, 4.521, 4.618,
    4.725, 4.841, 4.922, 5.004, 5.104, 5.196, 5.307, 5.418, 5.484, 5.549, 5.615, 5.681, 5.756, 5.842, 5.927, 6.023,
    6.119, 6.234, 6.360, 6.495, 6.651, 6.837, 7.072, 7.378, 7.864)
assert len(RATES) == 131 and len(LIMITS) == 131


def reference_ebno(rate):
    """`ebno(rate)` of simulation.c++:56-70, indexing included.

    rate <= 0.8 reads LIMITS[size_t(rate * 100)]: RATES[0] is 0.01, so this is the entry ONE rate step above
    the truncated rate (R = 16/31 = 0.516 reads the limit of R = 0.52).  rate >= 0.999 reads the last entry.
    In between, the first entry from index 80 (0.807) whose rate is >= `rate`, searched up to (not including)
    the last entry -- which is what the search returns when nothing matches.
    """
    rate = float(rate)
    if rate <= 0.800:
        return LIMITS[int(rate * 100)]
    if rate >= 0.999:
        return LIMITS[-1]
    index = 80
    while index < len(RATES) - 1 and not RATES[index] >= rate:
        index += 1
    return LIMITS[index]


def ladder(rate, step=0.5):
    """(start, max) of the Eb/N0 loop, simulation.c++:105-107: tmp = size_t(ebno(rate) / step) truncates toward
    zero (a negative limit gives tmp = 0: the conversion of a negative double is undefined in C++; every compiler
    the reference builds with yields 0 on x86-64 for values in (-1, 0], and the registry has no rate below 0.19)."""
    tmp = max(0, int(reference_ebno(rate) / step))
    start = (tmp + 1.0 / step) * step
    return start, max(8.0, start) + step / 2


def bpsk_capacity(snr_linear):
    """Capacity (bits/use) of the binary-input AWGN channel at Es/N0 = snr_linear, by Gauss-Hermite quadrature."""
    sigma2 = 1.0 / (2.0 * snr_linear)
    x, w = np.polynomial.hermite_e.hermegauss(96)
    y = 1.0 + math.sqrt(sigma2) * x
    llr = 2.0 * y / sigma2
    return 1.0 - float(np.sum(w * np.log2(1.0 + np.exp(-llr))) / math.sqrt(2.0 * math.pi))


def shannon_limit_ebno_db(rate):
    """Smallest Eb/N0 (dB) at which a rate-`rate` code can work on the BPSK-AWGN channel, solved numerically.
    Not used for the ladder (the reference's table is, `reference_ebno`); kept as the cross-check of that table
    (tests/test_host_logic.py: every entry within 0.12 dB of the solve)."""
    lo, hi = -3.0, 12.0
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if bpsk_capacity(rate * 10.0 ** (mid / 10.0)) >= rate:
            hi = mid
        else:
            lo = mid
    return hi


def samples(wer):
    """simulation.c++:91-93: min(1e6, 5e3 / wer); wer == 0 gives the cap (5e3 / 0.0 is +inf in the reference)."""
    return int(min(1e6, 5e3 / wer)) if wer > 0 else 1000000


def shard(total, rank, world):
    """Contiguous range of the global frame index owned by `rank`."""
    lo = total * rank // world
    hi = total * (rank + 1) // world
    return lo, hi - lo


class DeviceBackend:
    """Counts one shard of one Eb/N0 point on this rank's GPU through cc_mc_run_dev."""

    def __init__(self, code, random_codewords=False):
        import torch
        self.torch = torch
        self.code = code
        self.random_codewords = bool(random_codewords)
        self.device = torch.device("cuda", torch.cuda.current_device())

    def run(self, ebno_db, seed, first_frame, frames):
        torch = self.torch
        counters = torch.zeros(capi.MC_NCOUNTERS, dtype=torch.int64, device=self.device)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        rc = capi.lib().cc_mc_run_dev(self.code._h, float(ebno_db), int(seed), int(first_frame), int(frames),
                                      int(self.random_codewords), C.c_void_p(counters.data_ptr()), stream)
        capi.check(rc, "cc_mc_run_dev")
        return counters  # stays on the device: reduced with RCCL


class awgn_simulation:
    """awgn_simulation(decoder, step = 0.5, seed = 0) -- simulation.h:71-83."""

    def __init__(self, code, step=0.5, seed=0, random_codewords=False, backend=None, log_dir=None,
                 max_samples=None, start=None, stop=None, samples_per_point=None):
        self.code = code
        self.step = float(step)
        self.seed = int(seed)
        self.backend = backend if backend is not None else DeviceBackend(code, random_codewords)
        self.log_dir = log_dir
        self.max_samples = max_samples
        self.samples_per_point = samples_per_point  # fixed frame count per point instead of the adaptive rule
        ref_start, _ = ladder(code.rate, self.step)  # simulation.c++:105-107
        self.start = ref_start if start is None else float(start)
        self.stop = (max(8.0, self.start) + self.step / 2) if stop is None else float(stop)

    def points(self):
        e, out = self.start, []
        while e < self.stop:
            out.append(e)
            e += self.step
        return out

    def _dist(self):
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                return dist
        except ImportError:
            pass
        return None

    def run_point(self, ebno_db, frames, point_index=0):
        """Decode `frames` frames of one Eb/N0 point, sharded over the ranks; returns the reduced counters."""
        dist = self._dist()
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist else (0, 1)
        first, count = shard(frames, rank, world)
        # every point draws from its own stretch of the global frame sequence
        base = point_index << 40
        counters = self.backend.run(ebno_db, self.seed, base + first, count)
        if dist:
            dist.all_reduce(counters, op=dist.ReduceOp.SUM)  # the path's only exchange step
        c = counters.cpu().numpy() if hasattr(counters, "cpu") else np.asarray(counters)
        res = {k: int(c[i]) for k, i in COUNTER_NAMES.items()}
        res["iter_hist"] = [int(v) for v in c[capi.MC_ITER_HIST:]]
        res["ebno"] = ebno_db
        res["wer"] = res["word_errors"] / max(1, res["frames"])
        res["ber"] = res["bit_errors"] / max(1, res["frames"] * self.code.n)
        return res

    def _agree(self, ok, seed):
        """Rank 0's decision (log file opened or not) and its seed, made known to every rank BEFORE the first
        collective of the ladder: a rank that raised alone would leave the others blocked in the all-reduce."""
        dist = self._dist()
        if dist is None:
            return ok, seed
        import torch
        device = getattr(self.backend, "device", None)
        if device is None or dist.get_backend() != "nccl":
            device = "cpu"
        # the seed travels as two 31-bit halves + sign-free high part: int64 holds any 64-bit seed bit pattern
        t = torch.tensor([int(ok), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=torch.int64, device=device)
        dist.broadcast(t, src=0)
        v = t.cpu().tolist()
        return bool(v[0]), int(v[1]) | (int(v[2]) << 32)

    def __call__(self):
        """awgn_simulation::operator()(): the whole ladder; rank 0 writes the reference-format log."""
        dist = self._dist()
        rank = dist.get_rank() if dist else 0
        log, error = None, None
        if self.log_dir is not None and rank == 0:
            path = os.path.join(self.log_dir, self.code.to_string() + ".log")
            try:
                if os.path.exists(path):
                    raise RuntimeError("File %s already exists." % path)  # simulation.c++:72-81
                log = open(path, "w")
                log.write("%7s %21s\n" % ("ebno", "wer"))
            except (OSError, RuntimeError) as e:
                error = e
        ok, self.seed = self._agree(error is None, self.seed)
        if not ok:  # every rank leaves together
            raise error if error is not None else RuntimeError("rank 0 could not open the log file")
        wer, results = 0.5, []
        for idx, ebno in enumerate(self.points()):
            n = samples(wer) if self.samples_per_point is None else int(self.samples_per_point)
            if self.max_samples:
                n = min(n, self.max_samples)
            res = self.run_point(ebno, n, idx)
            results.append(res)
            wer = res["wer"]
            if log:
                log.write("%7s %s\n" % ("%.6g" % ebno, "%16.15e" % res["wer"]))
                log.flush()
        if log:
            log.close()
        return results


class bitflip_simulation:
    """bitflip_simulation(decoder, errors) -- simulation.h:85-92, simulation.c++:152-213.

    Exhaustive word-error rate by number of flipped bits: for every weight w <= errors and every one of the
    C(n, w) flip patterns the all-zero word is sent as +1 / -1 (x = -2*bit + 1, :190-191), decoded, and
    counted as a word error when the result is non-zero or the decoder fails (:193-199).  The reference
    walks the patterns with std::next_permutation one frame at a time; here all patterns of a weight are
    decoded as one batch on the GPU.
    """

    def __init__(self, code, errors=0, log_dir=None, batch=1 << 18):
        self.code, self.errors, self.log_dir, self.batch = code, int(errors), log_dir, int(batch)

    def patterns(self, w):
        """All C(n, w) flip patterns as rows of positions, in chunks of at most `batch` rows."""
        import itertools
        n = self.code.n
        if w == 0:
            yield np.zeros((1, 0), np.int64)
            return
        it = itertools.combinations(range(n), w)
        while True:
            chunk = np.fromiter(itertools.chain.from_iterable(itertools.islice(it, self.batch)), np.int64)
            if chunk.size == 0:
                return
            yield chunk.reshape(-1, w)

    def run_weight(self, w):
        import torch
        n = self.code.n
        patterns = word_errors = 0
        for pos in self.patterns(w):
            x = torch.ones((pos.shape[0], n), dtype=torch.float32, device="cuda")
            if w:
                x.scatter_(1, torch.from_numpy(pos).cuda(), -1.0)
            res = self.code.correct_batch(x)
            bad = (res["status"] != 0) | (res["out"] != 0).any(dim=1)
            patterns += pos.shape[0]
            word_errors += int(bad.sum())
        return patterns, word_errors

    def __call__(self):
        log = None
        if self.log_dir is not None:
            path = os.path.join(self.log_dir, self.code.to_string() + ".log")
            if os.path.exists(path):
                raise RuntimeError("File %s already exists." % path)
            log = open(path, "w")
            log.write("%7s %21s\n" % ("errors", "wer"))
        out = []
        for w in range(self.errors + 1):
            patterns, word_errors = self.run_weight(w)
            out.append(dict(errors=w, patterns=patterns, word_errors=word_errors, wer=word_errors / patterns))
            if log:
                log.write("%7s %s\n" % ("%.6g" % w, "%16.15e" % (word_errors / patterns)))
        if log:
            log.close()
        return out
